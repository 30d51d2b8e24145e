"""Row-band (multi-GPU) dense Lucas-Kanade against the single-device estimate (BASELINE config 5,
SURVEY 8e): 8 virtual ranks in lockstep on one device, collectives combined on the host, plus the
real RCCL collectives at world size 1.  Everything must be BIT-equal: statistics, both uint8
renderings inside the bands, the corner list with its order, the tracked vectors, the dense field."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(m, n, count, seed, nan=False):
    from oracle import semilag_cport as ocl
    from tools import synth

    base = synth.rain_field_db(m, n, seed=seed, sigma=max(m / 96.0, 2.0))
    vel = synth.true_velocity(m, n)
    adv = ocl.extrapolate(base, vel, count - 1, outval=-15.0)
    frames = np.stack([base] + [adv[t] for t in range(count - 1)])
    if nan:
        frames[:, synth.border_nan_mask(m, n, 0.1)] = np.nan
    return frames


@pytest.mark.parametrize("size,world,count,halo,nan,nr_levels", [
    (2048, 8, 2, 512, False, 3),   # the decomposition of config 5 at a quarter of its size
    (1024, 4, 3, 256, True, 3),    # NaN border (row 0 / 1 quirk, dilated mask), two pooled pairs
    (1024, 8, 2, 64, False, 3),    # halo far too small for the coarse levels: flagged tracks are redone
    (768, 3, 2, 512, False, 2),    # bands + halo cover the whole frame
])
def test_banded_lk_is_bit_equal_to_single_device(size, world, count, halo, nan, nr_levels):
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.motion import banded, get_method
    from pysteps_amd.motion import lucaskanade as lkmod

    m = n = size
    frames = _frames(m, n, count, seed=size + world, nan=nan)
    dframes = DeviceArray.from_host(frames)
    lk_kwargs = {"nr_levels": nr_levels}
    want = get_method("LK")(dframes, lk_kwargs=lk_kwargs).to_host()
    wxy, wuv = get_method("LK")(frames, dense=False, lk_kwargs=lk_kwargs)

    stats = [[] for _ in range(world)]
    gens = [banded.band_lucaskanade(dframes, r, world, halo=halo, nr_levels=nr_levels, stats_out=stats[r])
            for r in range(world)]
    fields = banded.run_virtual(gens)
    flagged = banded.band_lucaskanade.last_flagged
    for r in range(world):
        assert np.array_equal(fields[r].to_host(), want), "rank %d" % r
    # the statistics every rank ended up with are the single-device ones (first 5 slots: min, NaN
    # count, max, feature min / max)
    for t in range(count):
        prep = lkmod.PreparedFrame(dframes.view(t), 3, 5, True)
        single = prep.stats.to_host()
        for r in range(world):
            got = stats[r][t]
            assert np.array_equal(got[[0, 2, 3, 4]], single[[0, 2, 3, 4]]) and (got[1] > 0) == (single[1] > 0)
    if halo < 100:
        assert flagged > 0  # the fallback was exercised
    # sparse output of the banded path
    gens = [banded.band_lucaskanade(dframes, r, world, halo=halo, nr_levels=nr_levels, dense=False) for r in range(world)]
    sparse = banded.run_virtual(gens)
    for xy, uv in sparse:
        assert np.array_equal(xy, wxy) and np.array_equal(uv, wuv)


def test_banded_lk_through_rccl_at_world_size_one():
    """The same generator driven by the real communicator (ncclAllReduce / ncclAllGather on one rank)."""
    from pysteps_amd import parallel
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.motion import get_method

    frames = DeviceArray.from_host(_frames(512, 640, 2, seed=4))
    want = get_method("LK")(frames).to_host()
    comm = parallel.Communicator(0, 1, lambda payload: payload)
    try:
        assert np.array_equal(comm.allreduce_host(np.array([1.5, -2.0], np.float32), "min"), [1.5, -2.0])
        parts = comm.allgather_host(np.arange(6, dtype=np.uint64).reshape(3, 2))
        assert len(parts) == 1 and np.array_equal(parts[0], np.arange(6).reshape(3, 2))
        assert comm.allgather_host(np.empty((0, 4)))[0].shape == (0, 4)
        got = parallel.banded_dense_lucaskanade(frames, comm)
    finally:
        comm.close()
    assert np.array_equal(got.to_host(), want)
