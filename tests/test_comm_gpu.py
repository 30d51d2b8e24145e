"""RCCL communicator on one GPU (world_size 1): dlopen, ncclCommInitRank, broadcast, all-gather."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_roundtrip():
    from pysteps_amd import parallel
    from pysteps_amd.device import DeviceArray, synchronize

    comm = parallel.Communicator(0, 1, lambda payload: payload)
    try:
        a = np.arange(1 << 16, dtype=np.float32).reshape(2, 128, 256)
        d = DeviceArray.from_host(a)
        comm.broadcast(d, root=0)
        g = comm.allgather(d)
        synchronize()
        assert np.array_equal(d.to_host(), a)
        assert g.shape == (1, 2, 128, 256) and np.array_equal(g.to_host()[0], a)
    finally:
        comm.close()


def test_sharded_extrapolate_single_rank():
    from pysteps_amd import parallel
    from pysteps_amd.extrapolation import get_method
    from tools import synth

    m, n = 96, 128
    members = [synth.rain_field_db(m, n, seed=s) for s in range(3)]
    vel = synth.true_velocity(m, n)
    got = parallel.sharded_extrapolate(members, vel, 2, rank=0, world_size=1, outval=-15.0)
    assert sorted(got) == [0, 1, 2]
    want = get_method("semilagrangian")(members[1], vel, 2, outval=-15.0)
    assert np.array_equal(got[1], want)
