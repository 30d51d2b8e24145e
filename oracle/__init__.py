"""CPU oracles for the advection hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``pysteps_amd/`` imports this package; it is the checker used by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
"""
