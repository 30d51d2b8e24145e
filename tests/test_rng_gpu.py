"""numpy.random.RandomState.randn on the device (csrc/rng.hip, pysteps_amd.noise.randstate) against
NumPy itself - the generator the reference draws its white noise from (pysteps/noise/fftgenerators.py:400,
seeded by pysteps/nowcasts/steps.py:885-898).  Bar: the generator's state after every draw (624 key
words, position, cached value flag) identical with NumPy's; values identical except where glibc's log
is not correctly rounded (one ulp in the logarithm - up to 2^-52 relative - becomes at most 4 ulp in the value after the division, the root and the product, on less than
0.5 % of the values: tests/test_rng_cpu.py measures exactly that on the host); against the oracle
(oracle/randn.py: NumPy's own uniform stream + a correctly rounded log) every value is bit-identical."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ulp_diff(a, b):
    ia = a.view(np.int64)
    ib = b.view(np.int64)
    return np.abs(ia - ib)


def _check_values(got, want):
    assert got.shape == want.shape
    d = _ulp_diff(np.ascontiguousarray(got), np.ascontiguousarray(want))
    assert d.max() <= 4, "more than four ulp from NumPy's value"
    assert np.count_nonzero(d) <= max(2, 0.005 * d.size), "too many values differ from NumPy's"


def _same_state(a, b):
    assert a[0] == b[0]
    np.testing.assert_array_equal(a[1], b[1])
    assert a[2:4] == b[2:4]
    assert a[4] == b[4] or (a[3] == 0 and b[3] == 0)


def _steps_chain(seed, n):
    """the precipitation generators of nowcasts.steps (steps.py:885-898)"""
    out = []
    for _ in range(n):
        rs = np.random.RandomState(seed)
        out.append(rs)
        seed = rs.randint(0, high=int(1e9))
        seed = np.random.RandomState(seed).randint(0, high=int(1e9))
    return out


@pytest.mark.parametrize("shape", [(64, 64), (100, 37), (1, 1), (3,), (257, 129)])
def test_randn_bit_identical_with_the_oracle(shape):
    """device values == NumPy's uniform stream + correctly rounded log, bit for bit"""
    from oracle import randn as oracle

    from pysteps_amd.noise.randstate import DeviceRandomStates

    host = _steps_chain(1234, 2)
    host[1].standard_normal()  # starts with a cached value
    twin = [np.random.RandomState() for _ in host]
    for t, h in zip(twin, host):
        t.set_state(h.get_state())
    dev = DeviceRandomStates(twin, int(np.prod(shape)))
    for draw in range(2):
        got = dev.randn(*shape).to_host()
        for j, rs in enumerate(host):
            np.testing.assert_array_equal(got[j].ravel(), oracle.legacy_randn(rs, int(np.prod(shape))))
        for a, rs in zip(dev.get_states(), host):
            _same_state(a, rs.get_state())


@pytest.mark.parametrize("shape", [(64, 64), (100, 37), (512, 512), (1, 1), (3,), (1024, 1000)])
def test_randn_stream_matches_numpy(shape):
    from pysteps_amd.noise.randstate import DeviceRandomStates

    host = _steps_chain(42, 3)
    twin = [np.random.RandomState() for _ in host]
    for t, h in zip(twin, host):
        t.set_state(h.get_state())
    dev = DeviceRandomStates(twin, int(np.prod(shape)))
    for draw in range(3):
        got = dev.randn(*shape).to_host()
        for j, rs in enumerate(host):
            _check_values(got[j], rs.randn(*shape))
        for a, rs in zip(dev.get_states(), host):
            _same_state(a, rs.get_state())
    dev.sync_back()
    for t, rs in zip(twin, host):  # the host generators continue the same stream
        np.testing.assert_array_equal(t.randint(0, 1 << 30, 5), rs.randint(0, 1 << 30, 5))


def test_cached_value_and_odd_draws():
    """legacy_gauss keeps the second value of a pair: odd draws leave one cached, the next draw
    starts with it."""
    from pysteps_amd.noise.randstate import DeviceRandomStates

    host = [np.random.RandomState(s) for s in (1, 2, 3, 4)]
    host[1].standard_normal()  # one generator starts with a cached value
    host[2].random_sample(7)  # another in the middle of a block, position not a multiple of four
    twin = [np.random.RandomState() for _ in host]
    for t, h in zip(twin, host):
        t.set_state(h.get_state())
    dev = DeviceRandomStates(twin, 5001)
    for count in (5001, 1, 2, 4999, 5000, 3):
        got = dev.randn(count).to_host()
        for j, rs in enumerate(host):
            _check_values(got[j], rs.randn(count))
        for a, rs in zip(dev.get_states(), host):
            _same_state(a, rs.get_state())


def test_many_small_draws_resynchronise_the_ring():
    """the host's bounds on the stream positions drift apart by ~20 sigma per draw; small rings have
    to fall back on reading the true positions (rng_resync) and keep going"""
    from pysteps_amd.noise.randstate import DeviceRandomStates

    host = [np.random.RandomState(9), np.random.RandomState(10)]
    twin = [np.random.RandomState(9), np.random.RandomState(10)]
    dev = DeviceRandomStates(twin, 4096)
    for draw in range(120):
        got = dev.randn(64, 64)
        if draw % 40 == 39:
            got = got.to_host()
            want = None
            for j, rs in enumerate(host):
                want = rs.randn(64, 64)
                _check_values(got[j], want)
        else:
            for rs in host:
                rs.randn(64, 64)
    for a, rs in zip(dev.get_states(), host):
        _same_state(a, rs.get_state())


def test_side_stream_draw():
    from pysteps_amd.noise.randstate import DeviceRandomStates

    host = np.random.RandomState(5)
    dev = DeviceRandomStates([np.random.RandomState(5)], 256 * 256)
    a = dev.randn(256, 256, side=True)
    dev.wait()
    b = dev.randn(256, 256, side=True)
    dev.wait()
    _check_values(a.to_host()[0], host.randn(256, 256))
    _check_values(b.to_host()[0], host.randn(256, 256))


def test_full_size_field_4096():
    """one 4096^2 draw of two members: stream position and values at the size of BASELINE config 4"""
    from pysteps_amd.noise.randstate import DeviceRandomStates

    host = _steps_chain(7, 2)
    twin = [np.random.RandomState() for _ in host]
    for t, h in zip(twin, host):
        t.set_state(h.get_state())
    dev = DeviceRandomStates(twin, 4096 * 4096)
    got = dev.randn(4096, 4096)
    for j, rs in enumerate(host):
        _check_values(got.view(j).to_host(), rs.randn(4096, 4096))
    for a, rs in zip(dev.get_states(), host):
        _same_state(a, rs.get_state())


@pytest.mark.parametrize("shape,hint,draws", [((512, 512), 8, 4), ((512, 512), 1, 5), ((64, 64), 50, 6), ((1500, 1024), 2, 4)])
def test_chunked_production_by_jump_ahead(shape, hint, draws):
    """n_draws given: the streams are cut into chunks of 512 blocks whose start states come from the
    jump-ahead polynomials (mt_jump), every chunk produced by its own workgroup - the same words, so the
    same values and the same generator states as NumPy; when the hinted start states are used up the
    chunk grid is anchored anew at the last block produced and the stream goes on"""
    from pysteps_amd.noise.randstate import DeviceRandomStates

    host = _steps_chain(99, 3)
    host[2].random_sample(3)
    twin = [np.random.RandomState() for _ in host]
    for t, h in zip(twin, host):
        t.set_state(h.get_state())
    dev = DeviceRandomStates(twin, int(np.prod(shape)), n_draws=hint)
    for draw in range(draws):
        got = dev.randn(*shape, side=bool(draw & 1))
        dev.wait()
        got = got.to_host()
        for j, rs in enumerate(host):
            _check_values(got[j], rs.randn(*shape))
        for a, rs in zip(dev.get_states(), host):
            _same_state(a, rs.get_state())


@pytest.mark.parametrize("shape", [(64, 33), (101, 51), (1,), (1024, 513)])
def test_uniform_bit_identical_with_numpy(shape):
    """RandomState.uniform(low, high, size) - what the spectral-domain noise generator draws
    (pysteps/noise/fftgenerators.py:407): integer arithmetic on two words per value and one
    multiplication + addition, so the values are NumPy's bit for bit; draws of both kinds interleave on
    one stream (a cached normal value survives a uniform draw) and the host generators continue it."""
    from pysteps_amd.noise.randstate import DeviceRandomStates

    host = _steps_chain(7, 3)
    host[1].standard_normal()  # a cached value
    host[2].random_sample(5)
    twin = [np.random.RandomState() for _ in host]
    for t, h in zip(twin, host):
        t.set_state(h.get_state())
    count = int(np.prod(shape))
    dev = DeviceRandomStates(twin, count, n_draws=4)
    for draw in range(4):
        if draw == 2:
            got = dev.randn(*shape).to_host()
            for j, rs in enumerate(host):
                _check_values(got[j], rs.randn(*shape))
        else:
            got = dev.uniform(0.0, 2.0 * np.pi, *shape).to_host()
            for j, rs in enumerate(host):
                np.testing.assert_array_equal(got[j], rs.uniform(low=0.0, high=2.0 * np.pi, size=shape))
        for a, rs in zip(dev.get_states(), host):
            _same_state(a, rs.get_state())
    dev.sync_back()
    for t, rs in zip(twin, host):
        np.testing.assert_array_equal(t.randint(0, 1 << 30, 5), rs.randint(0, 1 << 30, 5))
