"""Dense Lucas-Kanade over row bands - the multi-GPU form of the motion estimate (BASELINE config 5:
8192 x 8192 tiled over 8 GPUs).

Every rank holds the whole input frames (one RCCL broadcast; 805 MB at 8192^2 is nothing next to
288 GB of HBM) and owns a band of rows.  What scales with the pixels is done on the band plus a halo,
what is global in the reference crosses ranks through three tiny collectives per frame:

===========================  ==========================================  =======================
stage (reference)             per rank                                    collective
===========================  ==========================================  =======================
min / NaN count of a frame    reduction over the own rows                 allreduce MIN, SUM
(lucaskanade.py:213-219)
opening + rescale ranges      band + halo; ranges over the own rows       allreduce MAX, MIN, MAX
(images.py:58-86,
tracking/lucaskanade.py:143-
160, shitomasi.py:143-151)
Shi-Tomasi response           band + halo; maximum over the own rows      allreduce MAX
(goodFeaturesToTrack)
corner candidates             3x3 maxima above 1 % of the maximum in      allgather (keys)
                              the own rows
min-distance selection        every rank: sort all keys, ordered greedy   -
                              pass (identical result everywhere)
pyramids + tracking           pyramid of band + halo, the corners that    allgather (vectors)
(calcOpticalFlowPyrLK)        lie in the own rows
outliers, declustering, IDW   every rank, whole field (the vector list    -
(cleansing.py, interpolate)   is tiny; the extrapolator needs the field
                              everywhere)
===========================  ==========================================  =======================

min / max reductions are order-free and the corner keys are unique, so every stage is bit-identical
to the single-device estimate.  A pyramid built from a band equals the whole-frame pyramid except
within a few rows of its inner edges; tracks whose windows reach those rows are flagged by the
kernel and redone on whole-frame data (never observed with the default 512-row halo).

The algorithm is written once, as a generator that yields its collectives
(``("allreduce", values, op)`` / ``("allgather", array)``); :func:`run` drives it with a real
:class:`pysteps_amd.parallel.Communicator`, :func:`run_virtual` drives several ranks in lockstep on
ONE device - that is how the 8-rank decomposition is tested on a single-GPU box.
"""

import ctypes

import numpy as np

from .. import _lib
from ..device import DeviceArray
from ..parallel import partition
from ..utils.cleansing import decluster, detect_outliers_device
from ..utils.interpolate import idw_to_device

__all__ = ["band_lucaskanade", "run", "run_virtual", "band_rows"]

_N_STATS = 8
_MIN_ALL, _NAN_COUNT, _MAX_ALL, _MIN_FEAT, _MAX_FEAT, _EIG_MAX = range(6)


def band_rows(m, world, rank, halo=512, align=16):
    """(r0, r1, e0, e1): own rows and the rows processed (own + halo; e0 aligned for the pyramid)."""
    own = partition(m, world, rank)
    r0, r1 = own.start, own.stop
    e0 = max(0, r0 - halo) // align * align
    e1 = min(m, r1 + halo)
    return r0, r1, e0, e1


class _BandFrame:
    """Device products of one frame on one rank (full-size buffers, band rows filled)."""

    def __init__(self, frame, rows, size_opening, buffer_mask, want_features):
        self.frame, self.rows = frame, rows
        self.m, self.n = frame.shape
        self.size_opening, self.buffer_mask = int(size_opening), int(buffer_mask)
        self.clean = DeviceArray((self.m, self.n), np.float32)
        self.track_u8 = DeviceArray((self.m, self.n), np.uint8)
        self.feature_u8 = DeviceArray((self.m, self.n), np.uint8) if want_features else None
        self.stats = DeviceArray((_N_STATS,), np.float32).fill_bytes(0)

    def prepare(self):
        """generator: stats -> opening -> uint8 renderings, with the three reductions in between"""
        lib = _lib.lib()
        r0, r1, e0, e1 = self.rows
        _lib.check(lib.psh_lk_band_stats_dev(self.frame.ptr, self.m, self.n, r0, r1, self.stats.ptr), "band_stats")
        s = self.stats.to_host()
        s[_MIN_ALL] = (yield ("allreduce", s[_MIN_ALL:_MIN_ALL + 1], "min"))[0]
        s[_NAN_COUNT] = (yield ("allreduce", s[_NAN_COUNT:_NAN_COUNT + 1], "sum"))[0]
        self._put_stats(s)
        _lib.check(lib.psh_lk_band_open_dev(self.frame.ptr, self.m, self.n, e0, e1, r0, r1, self.size_opening,
                                            self.buffer_mask, self.clean.ptr, self.stats.ptr), "band_open")
        s = self.stats.to_host()
        s[_MAX_ALL] = (yield ("allreduce", s[_MAX_ALL:_MAX_ALL + 1], "max"))[0]
        s[_MIN_FEAT] = (yield ("allreduce", s[_MIN_FEAT:_MIN_FEAT + 1], "min"))[0]
        s[_MAX_FEAT] = (yield ("allreduce", s[_MAX_FEAT:_MAX_FEAT + 1], "max"))[0]
        self._put_stats(s)
        self._to_u8(e0, e1)

    def _put_stats(self, s):
        self.global_stats = np.array(s, dtype=np.float32)
        _lib.check(_lib.lib().psh_memcpy_h2d(self.stats.ptr, self.global_stats.ctypes.data, self.global_stats.nbytes), "h2d")
        _lib.check(_lib.lib().psh_sync(), "sync")

    def _to_u8(self, e0, e1):
        _lib.check(_lib.lib().psh_lk_band_to_u8_dev(
            self.clean.ptr, self.m, self.n, e0, e1, self.buffer_mask, self.stats.ptr, self.track_u8.ptr,
            None if self.feature_u8 is None else self.feature_u8.ptr), "band_to_u8")

    def whole_frame(self):
        """Opening and tracker rendering of ALL rows with the global statistics (fallback for flagged
        tracks); the statistics block is untouched (the opening pass writes its own-row ranges into a
        scratch copy)."""
        lib = _lib.lib()
        r0, r1, _, _ = self.rows
        scratch = DeviceArray((_N_STATS,), np.float32)
        _lib.check(lib.psh_memcpy_d2d(scratch.ptr, self.stats.ptr, scratch.nbytes), "d2d")
        _lib.check(lib.psh_lk_band_open_dev(self.frame.ptr, self.m, self.n, 0, self.m, r0, r1, self.size_opening,
                                            self.buffer_mask, self.clean.ptr, scratch.ptr), "band_open")
        self._to_u8(0, self.m)
        self.rows = (r0, r1, 0, self.m)


def _corner_candidates(bf, block_size, quality_level):
    """generator: response (+ global maximum), candidates of the own rows -> uint64 keys (host)"""
    lib = _lib.lib()
    r0, r1, e0, e1 = bf.rows
    eig = DeviceArray((bf.m, bf.n), np.float32)
    _lib.check(lib.psh_lk_band_response_dev(bf.feature_u8.ptr, bf.clean.ptr, bf.m, bf.n, e0, e1, r0, r1, int(block_size),
                                            bf.buffer_mask, bf.stats.ptr, eig.ptr), "band_response")
    s = bf.stats.to_host()
    s[_EIG_MAX] = (yield ("allreduce", s[_EIG_MAX:_EIG_MAX + 1], "max"))[0]
    bf._put_stats(s)
    cap = max((r1 - r0) * bf.n // 6 + 4096, 4096)
    keys = DeviceArray((cap,), np.uint64)
    count = DeviceArray((1,), np.int32)
    _lib.check(lib.psh_lk_band_select_dev(eig.ptr, bf.clean.ptr, bf.m, bf.n, e0, e1, r0, r1, bf.buffer_mask,
                                          float(quality_level), bf.stats.ptr, keys.ptr, cap, count.ptr), "band_select")
    found = int(count.to_host()[0])
    if found > cap:
        raise RuntimeError("corner candidates of the band exceed the buffer (%d > %d)" % (found, cap))
    return keys.to_host()[:found].copy()


class _BandPyramid:
    def __init__(self, prev, nxt, winsize, nr_levels):
        lib = _lib.lib()
        if prev.rows != nxt.rows:  # one of the two was completed for an earlier fallback: complete the other
            for f in (prev, nxt):
                if f.rows[2:] != (0, f.m):
                    f.whole_frame()
        r0, r1, e0, e1 = prev.rows
        self.handle = ctypes.c_void_p()
        off = e0 * prev.n
        _lib.check(lib.psh_lk_pyramids_dev(prev.track_u8.ptr + off, nxt.track_u8.ptr + off, e1 - e0, prev.n,
                                           int(winsize[0]), int(winsize[1]), int(nr_levels), ctypes.byref(self.handle)),
                   "psh_lk_pyramids_dev")
        self.whole = e0 == 0 and e1 == prev.m
        self.usable = True
        if not self.whole:
            rc = lib.psh_lk_pyramids_band(self.handle, prev.m, e0, None)
            if rc == _lib.PSH_EUNSUPPORTED:
                self.usable = False  # band too small for the frame's level count: whole-frame tracking
            else:
                _lib.check(rc, "psh_lk_pyramids_band")
        self._keep = (prev, nxt)

    def track(self, points, criteria, min_eig_thr):
        from .lucaskanade import _criteria  # noqa: PLC0415

        p0 = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 2)
        p1 = np.empty_like(p0)
        st = np.zeros(p0.shape[0], dtype=np.uint8)
        if p0.shape[0]:
            max_count, eps = _criteria(criteria)
            _lib.check(_lib.lib().psh_lk_track_pyr_dev(self.handle, p0.ctypes.data, p0.shape[0], max_count, eps,
                                                       float(min_eig_thr), p1.ctypes.data, st.ctypes.data),
                       "psh_lk_track_pyr_dev")
        return p1, st

    def close(self):
        if self.handle:
            _lib.load().psh_lk_pyramids_free(self.handle)
            self.handle = None


def band_lucaskanade(frames, rank, world, halo=512, size_opening=3, buffer_mask=5, max_corners=1000,
                     quality_level=0.01, min_distance=10, block_size=5, winsize=(50, 50), nr_levels=3,
                     criteria=(3, 10, 0), min_eig_thr=1e-4, nr_std_outlier=3, k_outlier=30, decl_scale=20,
                     interp_kwargs=None, dense=True, stats_out=None):
    """Generator form of ``dense_lucaskanade`` for rank ``rank`` of ``world`` (see module docstring).
    ``frames``: float32 DeviceArray ``(T, m, n)`` present on every rank.  Returns (StopIteration
    value) the float32 DeviceArray ``(2, m, n)`` - the whole field, on every rank - or ``(xy, uv)``."""
    if not isinstance(frames, DeviceArray) or frames.dtype != np.float32 or frames.ndim != 3:
        raise ValueError("band_lucaskanade works on a float32 DeviceArray (T, m, n)")
    interp_kwargs = dict(interp_kwargs or {})
    nr_fields, m, n = frames.shape
    rows = band_rows(m, world, rank, halo, 1 << max(int(nr_levels), 0))
    r0, r1 = rows[:2]
    prepared = []
    for t in range(nr_fields):
        bf = _BandFrame(frames.view(t), rows, size_opening, buffer_mask, want_features=t < nr_fields - 1)
        yield from bf.prepare()
        prepared.append(bf)
    if stats_out is not None:
        stats_out.extend(bf.global_stats.copy() for bf in prepared)

    xy = np.empty((0, 2))
    uv = np.empty((0, 2))
    flagged_total = 0
    for t in range(nr_fields - 1):
        mine = yield from _corner_candidates(prepared[t], block_size, quality_level)
        gathered = yield ("allgather", mine)
        keys = np.sort(np.concatenate(gathered))[::-1].copy()  # descending: strongest first, ties by address
        pts = np.empty((int(max_corners), 2), dtype=np.float32)
        count = ctypes.c_int(0)
        _lib.check(_lib.lib().psh_lk_greedy_host(keys.ctypes.data, int(keys.size), m, n, float(min_distance),
                                                 int(max_corners), pts.ctypes.data, ctypes.byref(count)), "lk_greedy")
        pts = pts[: count.value].copy()
        own = np.flatnonzero((pts[:, 1] >= r0) & (pts[:, 1] < r1)) if len(pts) else np.empty(0, np.int64)
        p1 = np.empty((len(own), 2), np.float32)
        st = np.zeros(len(own), np.uint8)
        if len(own):
            pyr = _BandPyramid(prepared[t], prepared[t + 1], winsize, nr_levels)
            if pyr.usable:
                p1, st = pyr.track(pts[own], criteria, min_eig_thr)
            else:
                st[:] = 2
            pyr.close()
            redo = np.flatnonzero(st & 2)
            if len(redo):  # windows that reached rows a band pyramid does not reproduce: whole-frame data
                flagged_total += len(redo)
                prepared[t].whole_frame()
                prepared[t + 1].whole_frame()
                full = _BandPyramid(prepared[t], prepared[t + 1], winsize, nr_levels)
                p1r, str_ = full.track(pts[own][redo], criteria, min_eig_thr)
                full.close()
                p1[redo], st[redo] = p1r, str_
        packed = np.concatenate([own.astype(np.float64)[:, None], p1.astype(np.float64),
                                 st.astype(np.float64)[:, None]], axis=1)
        parts = yield ("allgather", packed)
        allp = np.concatenate(parts) if parts else np.empty((0, 4))
        order = np.argsort(allp[:, 0], kind="stable")  # back to corner order
        allp = allp[order]
        if len(allp) != len(pts):
            raise RuntimeError("band ownership of the corners is not a partition (%d of %d)" % (len(allp), len(pts)))
        ok = allp[:, 3] == 1
        next_pts = allp[:, 1:3].astype(np.float32)
        if ok.any():
            xy = np.append(xy, pts[ok], axis=0)
            uv = np.append(uv, next_pts[ok] - pts[ok], axis=0)
    band_lucaskanade.last_flagged = flagged_total

    def zero_field():
        return DeviceArray((2, m, n), np.float32).fill_bytes(0)

    if xy.shape[0] == 0:
        return zero_field() if dense else (xy, uv)
    outliers = detect_outliers_device(uv, nr_std_outlier, xy, k_outlier, False)
    xy, uv = xy[~outliers, :], uv[~outliers, :]
    if not dense:
        return xy, uv
    if decl_scale > 1:
        xy, uv = decluster(xy, uv, decl_scale, 1, False)
    if xy.shape[0] == 0:
        return zero_field()
    if xy.shape[0] == 1 or uv.max() == uv.min():
        return DeviceArray.from_host(np.ones((2, m, n), dtype=np.float32) * uv[0].astype(np.float32)[:, None, None])
    return idw_to_device(xy, uv, m, n, power=interp_kwargs.get("power", 0.5), k=interp_kwargs.get("k", 20),
                         dist_offset=interp_kwargs.get("dist_offset", 0.5))


_REDUCERS = {"min": np.minimum.reduce, "max": np.maximum.reduce, "sum": np.add.reduce}


def run_virtual(generators):
    """Drive the generators of several ranks in lockstep on one device; collectives are combined on
    the host.  Returns the list of results (one per rank)."""
    gens = list(generators)
    results = [None] * len(gens)
    pending = [None] * len(gens)
    live = [True] * len(gens)
    for i, g in enumerate(gens):
        try:
            pending[i] = next(g)
        except StopIteration as stop:
            live[i], results[i] = False, stop.value
    while any(live):
        if not all(live):
            raise RuntimeError("ranks left the collective sequence at different points")
        kinds = {p[0] for p in pending}
        if len(kinds) != 1:
            raise RuntimeError("ranks disagree on the collective: %s" % sorted(kinds))
        kind = kinds.pop()
        if kind == "allreduce":
            ops = {p[2] for p in pending}
            if len(ops) != 1:
                raise RuntimeError("ranks disagree on the reduction")
            answer = _REDUCERS[ops.pop()]([np.asarray(p[1], dtype=np.float32) for p in pending]).astype(np.float32)
            answers = [answer.copy() for _ in gens]
        else:
            gathered = [np.array(p[1], copy=True) for p in pending]
            answers = [list(gathered) for _ in gens]
        for i, g in enumerate(gens):
            try:
                pending[i] = g.send(answers[i])
            except StopIteration as stop:
                live[i], results[i] = False, stop.value
    return results


def run(generator, comm):
    """Drive one rank's generator with a real communicator (:class:`pysteps_amd.parallel.Communicator`)."""
    try:
        req = next(generator)
        while True:
            if req[0] == "allreduce":
                answer = comm.allreduce_host(np.asarray(req[1], dtype=np.float32), req[2])
            else:
                answer = comm.allgather_host(np.asarray(req[1]))
            req = generator.send(answer)
    except StopIteration as stop:
        return stop.value
