"""Host-side pin of csrc/cr_log.h, the double-double logarithm of the device random stream
(csrc/rng.hip): the same header compiled by g++ must return the correctly rounded natural logarithm
- checked against decimal at 60 digits - on the arguments the polar method produces (0 < r2 < 1), and
the numbers DESIGN.md quotes for the C library's own log() are measured here."""

import ctypes
import math
import os
import subprocess
from decimal import Decimal, getcontext

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = r"""
#include "%s/pysteps_amd/csrc/cr_log.h"
static const double T[PSH_CRLOG_N][3] = {PSH_CRLOG_TABLE};
extern "C" void crlog_array(const double *x, double *y, long n) {
  for (long i = 0; i < n; ++i) y[i] = psh::crlog::log_cr(x[i], T);
}
"""


@pytest.fixture(scope="module")
def crlog(tmp_path_factory):
    d = tmp_path_factory.mktemp("crlog")
    src, so = d / "shim.cpp", d / "shim.so"
    src.write_text(SHIM % ROOT)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))

    def fn(x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        lib.crlog_array(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size))
        return y

    return fn


def _exact(x):
    getcontext().prec = 60
    return np.array([float(Decimal(float(v)).ln()) for v in x])  # Decimal -> float rounds correctly


def test_log_is_correctly_rounded(crlog):
    rng = np.random.default_rng(7)
    cases = [
        rng.random(40000),
        1.0 - rng.random(20000) * 2.0 ** -rng.integers(1, 40, 20000),  # just below one: no cancellation allowed
        rng.random(10000) * 2.0 ** -rng.integers(1, 104, 10000),  # down to the smallest r2 two 2^-52 steps give
        np.array([0.75, 0.5, 0.25, 1 - 2.0 ** -53, 1 - 2.0 ** -9, 1 - 2.0 ** -9 - 2.0 ** -53, 0.7499999999999999,
                  2.0 ** -104, 0.99609375, 0.998046875, 1.5 / 2 + 2.0 ** -53]),
    ]
    glibc_off = total = 0
    for x in cases:
        x = x[(x > 0) & (x < 1)]
        want = _exact(x)
        np.testing.assert_array_equal(crlog(x), want)
        glibc_off += int(np.count_nonzero(np.array([math.log(v) for v in x]) != want))
        total += x.size
    # the C library's log is faithful, not correctly rounded: that is why device and NumPy values may
    # differ by one ulp; the rate seen with glibc 2.35 is ~1e-3
    assert glibc_off < 0.01 * total


def test_table_is_what_the_generator_writes(tmp_path):
    """cr_log_table.h is generated (tools/gen_log_table.py); the committed file must be its output."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_log_table", os.path.join(ROOT, "tools", "gen_log_table.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(gen.OUT).read()
    gen.OUT = str(tmp_path / "t.h")
    gen.main()
    assert open(gen.OUT).read() == committed


def test_oracle_restatement_reproduces_numpy():
    """oracle/randn.py with the C library's log IS RandomState.randn: values and generator state"""
    from oracle import randn as oracle

    for seed, counts in ((1, (10, 7, 1, 4096)), (2, (1, 1, 2, 3)), (3, (100001,))):
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        a.random_sample(3)
        b.random_sample(3)
        for count in counts:
            want = a.randn(count)
            # math.log is the C library's scalar log legacy_gauss calls; numpy's array log is a SIMD
            # implementation of its own and rounds differently
            got = oracle.legacy_randn(b, count, log=lambda x: np.array([math.log(v) for v in x]))
            np.testing.assert_array_equal(got, want)
            sa, sb = a.get_state(), b.get_state()
            np.testing.assert_array_equal(sa[1], sb[1])
            assert sa[2:] == sb[2:]


def test_correct_rounding_moves_few_values_by_few_ulps():
    """what replacing glibc's log by the correctly rounded one does to the stream (the tolerance of
    tests/test_rng_gpu.py): < 0.5 % of the values, a few ulp (one ulp of the logarithm is up to 2^-52 relative)"""
    from oracle import randn as oracle

    a, b = np.random.RandomState(11), np.random.RandomState(11)
    want = a.randn(60000)
    got = oracle.legacy_randn(b, 60000)
    d = np.abs(got.view(np.int64) - want.view(np.int64))
    assert d.max() <= 4
    assert np.count_nonzero(d) < 0.005 * d.size
    assert a.get_state()[2:] == b.get_state()[2:]


def _jump_table():
    import re

    text = open(os.path.join(ROOT, "pysteps_amd", "csrc", "mt_jump_tables.h")).read()
    chunk_blocks = int(re.search(r"PSH_MT_CHUNK_BLOCKS (\d+)", text).group(1))
    rows = re.findall(r"\{([^}]*)\}", text[text.index("PSH_MT_JUMP_TABLE"):])
    table = np.array([[int(v.strip().rstrip("u"), 16) for v in row.split(",")] for row in rows], dtype=np.uint64)
    return chunk_blocks, table


def _mix(a, b):
    y = (a & np.uint64(0x80000000)) | (b & np.uint64(0x7FFFFFFF))
    return (y >> np.uint64(1)) ^ np.where(b & np.uint64(1), np.uint64(0x9908B0DF), np.uint64(0))


def _horner32(poly_words, window):
    """g(A) window the way csrc/rng.hip `mt_jump` evaluates it: 32 coefficients per round, the window as a
    ring whose 32 dropped words are overwritten by the 32 new ones"""
    n = 624
    ext = np.concatenate([window, np.zeros(32, np.uint64)])
    t = np.arange(31)
    ext[n + t] = ext[t + 397] ^ _mix(ext[t], ext[t + 1])
    acc = np.zeros(n, np.uint64)
    idx = np.arange(n)
    for c in range(n - 1, -1, -1):
        chunk = int(poly_words[c])
        j = np.arange(32)
        fresh = acc[j + 397] ^ _mix(acc[j], acc[j + 1])  # logical indexing: acc is kept in logical order here
        x = np.zeros(n, np.uint64)
        while chunk:
            b = (chunk & -chunk).bit_length() - 1
            x ^= ext[idx + b]
            chunk &= chunk - 1
        acc = np.concatenate([acc[32:], fresh]) ^ x
    return acc


def test_jump_polynomials_move_a_numpy_generator_by_whole_chunks():
    """mt_jump_tables.h (tools/gen_mt_jump.py): level l jumps a state 624 * 512 * 2^l words ahead - checked
    against NumPy drawing that many words, with the 32-coefficients-per-round Horner rule of the kernel"""
    chunk_blocks, table = _jump_table()
    assert table.shape[1] == 624 and table.shape[0] >= 10
    rs = np.random.RandomState(2024)
    rs.random_sample(5)  # somewhere inside a block: the key array is what jumps, not the position
    key = rs.get_state()[1].astype(np.uint64)
    for level in (0, 1, 3):
        words = 624 * chunk_blocks << level
        twin = np.random.RandomState()
        twin.set_state(("MT19937", key.astype(np.uint32), 624, 0, 0.0))
        twin.randint(0, 1 << 32, size=words, dtype=np.uint32)  # exactly one word each: the key is now `words` ahead
        want = twin.get_state()[1].astype(np.uint64)
        got = _horner32(table[level], key)
        np.testing.assert_array_equal(got[1:], want[1:])
        assert (int(got[0]) ^ int(want[0])) >> 31 == 0  # the low 31 bits of the first word are never read


def test_jump_table_is_what_the_generator_writes(tmp_path):
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_mt_jump", os.path.join(ROOT, "tools", "gen_mt_jump.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(gen.OUT).read()
    gen.OUT = str(tmp_path / "t.h")
    gen.main()
    assert open(gen.OUT).read() == committed
