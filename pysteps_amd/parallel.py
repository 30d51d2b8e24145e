"""Multi-GPU plumbing: one process (rank) per GPU, RCCL over xGMI.

The advection path shards embarrassingly (ensemble members, independent fields or
lead-time batches, output tiles): every rank advects its own share and the only
data-path collective is one broadcast of the input fields from the rank that
produced them (``Communicator.broadcast``).  The reference has nothing to mirror
here - its only parallelism is dask threads over members
(pysteps/nowcasts/utils.py:464-471, nowcasts/steps.py:705-720).
"""

import ctypes

import numpy as np

from . import _lib
from .device import DeviceArray


def partition(n_items, world_size, rank):
    """Contiguous share of ``n_items`` for ``rank`` (members j -> rank j // ceil(n/world);
    48 members on 8 GPUs -> 6 each).  Returns a ``range``."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank %r of %r" % (rank, world_size))
    if n_items < 0:
        raise ValueError("n_items must be >= 0")
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def owner_of(item, n_items, world_size):
    """Rank that owns ``item`` under :func:`partition`."""
    if not (0 <= item < n_items):
        raise ValueError("item out of range")
    base, extra = divmod(n_items, world_size)
    boundary = extra * (base + 1)
    if item < boundary:
        return item // (base + 1)
    return extra + (item - boundary) // base


class Communicator:
    """RCCL communicator on the GPU this process is bound to.

    ``exchange(payload_or_None) -> payload`` must broadcast a small bytes object
    from rank 0 to all ranks over any host channel (torch.distributed object
    broadcast, MPI, a file, ...); it carries the 128-byte ncclUniqueId.
    """

    def __init__(self, rank, world_size, exchange):
        self.rank, self.world_size = int(rank), int(world_size)
        lib = _lib.lib()
        nbytes = lib.psh_comm_unique_id_bytes()
        uid = None
        if self.rank == 0:
            buf = ctypes.create_string_buffer(nbytes)
            _lib.check(lib.psh_comm_unique_id(buf), "psh_comm_unique_id")
            uid = buf.raw
        uid = exchange(uid)
        if not isinstance(uid, (bytes, bytearray)) or len(uid) != nbytes:
            raise ValueError("exchange() must return the %d-byte unique id of rank 0" % nbytes)
        _lib.check(lib.psh_comm_init(bytes(uid), self.world_size, self.rank), "psh_comm_init")

    def broadcast(self, array, root=0):
        """In-place broadcast of a DeviceArray (asynchronous on the library stream)."""
        if not isinstance(array, DeviceArray):
            raise TypeError("broadcast expects a DeviceArray")
        _lib.check(_lib.lib().psh_comm_broadcast(array.ptr, array.nbytes, int(root)), "psh_comm_broadcast")
        return array

    def allgather(self, array):
        """Gather equally sized DeviceArrays from all ranks -> (world_size,) + shape."""
        out = DeviceArray((self.world_size,) + array.shape, array.dtype)
        _lib.check(_lib.lib().psh_comm_allgather(array.ptr, out.ptr, array.nbytes), "psh_comm_allgather")
        return out

    _OPS = {"sum": 0, "max": 1, "min": 2}  # PSH_COMM_* of include/pysteps_hip.h

    def allreduce(self, array, op):
        """In-place ncclAllReduce of a float32 DeviceArray (``op``: "min", "max" or "sum")."""
        if not isinstance(array, DeviceArray) or array.dtype != np.float32:
            raise TypeError("allreduce expects a float32 DeviceArray")
        _lib.check(_lib.lib().psh_comm_allreduce_f32(array.ptr, array.size, self._OPS[op]), "psh_comm_allreduce_f32")
        return array

    def allreduce_host(self, values, op):
        """A few float32 values from the host through the device collective and back (the global
        statistics of the row-band Lucas-Kanade passes, ``motion/banded.py``)."""
        buf = DeviceArray.from_host(np.ascontiguousarray(values, dtype=np.float32))
        return self.allreduce(buf, op).to_host()

    def allgather_host(self, array):
        """Gather host arrays whose leading dimension differs between ranks -> list in rank order
        (corner candidates, tracked vectors: kilobytes).  Two device collectives: the lengths, then
        the payloads padded to the longest."""
        arr = np.ascontiguousarray(array)
        lengths = self.allgather(DeviceArray.from_host(np.array([arr.shape[0]], dtype=np.int64))).to_host().ravel()
        longest = int(lengths.max())
        if longest == 0:
            return [arr[:0].copy() for _ in range(self.world_size)]
        padded = np.zeros((longest,) + arr.shape[1:], dtype=arr.dtype)
        padded[: arr.shape[0]] = arr
        allp = self.allgather(DeviceArray.from_host(padded)).to_host()
        return [allp[r, : int(lengths[r])].copy() for r in range(self.world_size)]

    def close(self):
        _lib.check(_lib.lib().psh_comm_destroy(), "psh_comm_destroy")


def steps_shard(seed, n_ens_members, world_size, rank):
    """The share of a ``pysteps.nowcasts.steps`` ensemble that ``rank`` runs (BASELINE config 4: 48 members,
    6 per GPU) -> ``(members, kwargs)``: the global member indices and the ``n_ens_members`` / ``seed``
    keywords to hand to ``nowcasts.steps`` on this rank.

    The nowcaster seeds its members from ONE chain (pysteps/nowcasts/steps.py:885-898): per member a
    generator for the precipitation noise, ``seed = rs.randint(0, 1e9)``, a generator for the motion
    perturbation, ``seed = rs.randint(0, 1e9)``.  Walking the chain to this rank's first member gives the
    seed from which the nowcaster, run with ``n_ens_members=len(members)``, builds exactly the generators
    of the global members ``members`` - so N ranks produce the members of the single-process ensemble,
    every one on its rank, with no communication beyond the broadcast of the inputs.  ``seed=None``
    (unseeded run) stays None."""
    import numpy.random as npr  # noqa: PLC0415

    members = partition(n_ens_members, world_size, rank)
    if seed is not None:
        for _ in range(members.start):
            seed = npr.RandomState(seed).randint(0, high=int(1e9))
            seed = npr.RandomState(seed).randint(0, high=int(1e9))
    return members, {"n_ens_members": len(members), "seed": seed}


def sharded_extrapolate(precip_members, velocity, timesteps, rank, world_size, **kwargs):
    """Advect this rank's share of an ensemble (list of (m,n) fields, same velocity).

    Inputs must already be present on every rank (see :meth:`Communicator.broadcast`);
    returns ``{member_index: result}`` for the members this rank owns.  No
    communication happens here - members are independent.
    """
    from .extrapolation import get_method

    extrapolate = get_method("semilagrangian")
    mine = partition(len(precip_members), world_size, rank)
    return {j: extrapolate(precip_members[j], velocity, timesteps, **kwargs) for j in mine}


def banded_dense_lucaskanade(frames, comm, **kwargs):
    """Dense Lucas-Kanade of frames that every rank holds, the image passes tiled into row bands
    over the ranks of ``comm`` (BASELINE config 5; see :mod:`pysteps_amd.motion.banded`).  Returns
    the whole (2,m,n) field as a DeviceArray on every rank - the form ``tiled_extrapolate`` takes."""
    from .motion import banded

    return banded.run(banded.band_lucaskanade(frames, comm.rank, comm.world_size, **kwargs), comm)


def config5_plan(m, world_size, rank, lk="replicated"):
    """What ``rank`` does in one step of the row-band nowcast (BASELINE config 5) - no device needed.

    ``lk="replicated"`` (the default): every rank runs the WHOLE dense Lucas-Kanade estimate on the frames it holds.
    The estimate is deterministic (integer image passes, ordered corner walk, fixed-order k-NN sums), so all ranks
    arrive at the same field bit for bit and the step has NO data-path collective: the estimate costs 1.5 ms at
    8192^2 on one MI355X, less than moving its 512 MiB result over xGMI would (>= 3 ms at the links' 153 GB/s), and a
    row-band estimate cannot pay either - its sequential stages (corner walk, vector QC) do not shrink with the band
    and each needs a collective (``lk="banded"``: 3 small allreduces + 2 allgathers per estimate,
    :mod:`pysteps_amd.motion.banded`).  Only the extrapolation (5.5 ms at 8192^2 x 36 lead times) is worth tiling:
    expected time per rank ~ 1.5 + 5.5 / world_size ms."""
    if lk not in ("replicated", "banded"):
        raise ValueError("lk must be 'replicated' or 'banded'")
    rows = partition(m, world_size, rank)
    return {"rows": (rows.start, rows.stop), "lk": lk,
            "collectives_per_step": 0 if lk == "replicated" else 5,
            "expected_ms_8192": 1.5 + 5.5 / world_size if lk == "replicated" else None}


def config5_step(frames, timesteps, rank, world_size, comm=None, lk="replicated", lk_kwargs=None, **kwargs):
    """One step of the row-band nowcast on frames every rank holds (one broadcast, outside the step): the motion
    field (see :func:`config5_plan` for the two ways), then this rank's row band of the extrapolation of the last
    frame.  Returns ``(rows, out)`` as :func:`tiled_extrapolate` does; the bands of all ranks concatenate to the
    single-device result bit for bit (tests/test_semilag_gpu.py, tests/test_lk_banded_gpu.py)."""
    from .motion import get_method

    plan = config5_plan(frames.shape[1], world_size, rank, lk)
    if plan["lk"] == "banded":
        if comm is None:
            raise ValueError("lk='banded' needs the communicator its collectives run on")
        v = banded_dense_lucaskanade(frames, comm, **(lk_kwargs or {}))
    else:
        v = get_method("LK")(frames, **(lk_kwargs or {}))
    return tiled_extrapolate(frames.view(frames.shape[0] - 1), v, timesteps, rank, world_size, **kwargs)


def tiled_extrapolate(precip, velocity, timesteps, rank, world_size, outval=float("nan"), n_iter=1,
                      interp_order=1):
    """Output-tiled nowcast for domains shared by several GPUs (BASELINE config 5).

    Every rank holds the whole (broadcast) ``precip`` (m,n) and ``velocity`` (2,m,n)
    DeviceArrays - they are tiny next to 288 GB of HBM - and integrates only the pixels of
    its row band ``partition(m, world_size, rank)``; the semi-Lagrangian scheme has no
    inter-pixel dependency, so no halo exchange is needed and the bands of all ranks
    concatenate to the single-GPU result bit for bit.  Returns ``(rows, out)`` with
    ``out`` a float32 DeviceArray ``(T, len(rows), n)``.
    """
    from .extrapolation.semilagrangian import _step_increments

    if not isinstance(precip, DeviceArray) or not isinstance(velocity, DeviceArray):
        raise TypeError("tiled_extrapolate works on device-resident fields")
    m, n = precip.shape
    rows = partition(m, world_size, rank)
    steps = _step_increments(timesteps, 1)
    out = DeviceArray((steps.size, max(len(rows), 1), n), np.float32)
    if len(rows) == 0:
        return rows, out
    rc = _lib.lib().psh_semilag_rows_dev(
        precip.ptr, velocity.ptr, m, n, steps.ctypes.data, int(steps.size), int(n_iter), int(interp_order),
        float(outval), None, 0, rows.start, len(rows), out.ptr,
    )
    _lib.check(rc, "psh_semilag_rows_dev")
    return rows, out
