"""Bit-level fingerprints of the dense Lucas-Kanade stages over a spread of frames (development
aid, the LK twin of tools/sl_bitcheck.py): run before and after a kernel change; the digests of the
stages that change said to leave alone must not move.

    python tools/lk_bitcheck.py <tag>      -> gpurun_out/lk_bitcheck_<tag>.json
    python tools/lk_bitcheck.py --diff a b

Per case: cleaned frame, both uint8 renderings, statistics block, Shi-Tomasi response (through the
row-band entry point with the whole frame as the band), accepted corners, tracked points + status,
sparse vectors and dense field of the whole estimate, and an IDW field from a fixed vector set.
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, ".")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def frames_for(m, n, count, seed, nan):
    from pysteps_amd.extrapolation import get_method
    from tools import synth

    base = synth.rain_field_db(m, n, seed=seed)
    vel = synth.true_velocity(m, n)
    if count > 1:
        adv = get_method("semilagrangian")(base, vel, count - 1, outval=float(base.min()))
        frames = np.concatenate([base[None], adv]).astype(np.float32)
    else:
        frames = base[None].copy()
    if nan:
        frames[:, synth.border_nan_mask(m, n)] = np.nan
        frames[:, m // 3:m // 3 + 7, n // 2:n // 2 + 11] = np.nan
    return frames


def run(tag):
    from pysteps_amd import _lib
    from pysteps_amd.device import DeviceArray, synchronize
    from pysteps_amd.motion import lucaskanade as lk
    from pysteps_amd.utils.interpolate import idw_to_device

    lib = _lib.lib()
    res = {}
    cases = [(257, 389, 2, 5, True, np.float32), (512, 512, 2, 7, False, np.float32), (1024, 1000, 3, 11, True, np.float64),
             (2048, 2048, 2, 13, False, np.float32), (4096, 4096, 2, 1234, False, np.float32), (130, 1031, 2, 3, True, np.float32)]
    for (m, n, count, seed, nan, dtype) in cases:
        key = "%dx%d f%d %s%s" % (m, n, count, np.dtype(dtype).name, " nan" if nan else "")
        frames = frames_for(m, n, count, seed, nan).astype(dtype)
        fd = DeviceArray.from_host(frames)
        preps = [lk.PreparedFrame(fd.view(t), 3, 5, t < count - 1) for t in range(count)]
        r = {}
        for t, p in enumerate(preps):
            r["clean%d" % t] = digest(p.clean.to_host())
            r["trk%d" % t] = digest(p.track_u8.to_host())
            if p.feature_u8 is not None:
                r["feat%d" % t] = digest(p.feature_u8.to_host())
            r["stats%d" % t] = digest(p.stats.to_host()[:5])
        p0, p1 = preps[0], preps[1]
        eig = DeviceArray((m, n), np.float32)
        stats_copy = DeviceArray((8,), np.float32)
        _lib.check(lib.psh_memcpy_d2d(stats_copy.ptr, p0.stats.ptr, 32))
        _lib.check(lib.psh_lk_band_response_dev(p0.feature_u8.ptr, p0.clean.ptr, m, n, 0, m, 0, m, 5, 5, stats_copy.ptr,
                                                eig.ptr), "band_response")
        r["eig"] = digest(eig.to_host())
        r["eigmax"] = digest(stats_copy.to_host()[5:6])
        for (mc, md) in ((1000, 10.0), (300, 25.0), (2500, 6.5)):
            pts = lk.detect_corners(p0, max_corners=mc, min_distance=md)
            r["corners %d %g" % (mc, md)] = [digest(pts), len(pts)]
        pts = lk.detect_corners(p0)
        for win in ((50, 50), (21, 31)):
            nxt, st = lk.track_points(p0, p1, pts, winsize=win)
            r["track %dx%d" % win] = [digest(nxt[st]), digest(st), int(st.sum())]
        xy, uv = lk.dense_lucaskanade(frames, dense=False)
        r["sparse"] = [digest(xy), digest(uv), len(xy)]
        field = lk.dense_lucaskanade(frames)
        r["dense"] = digest(field)
        r["dense32"] = digest(lk.dense_lucaskanade(DeviceArray.from_host(frames.astype(np.float32))).to_host())
        # IDW from a fixed vector set (independent of the front end): k = 20 and k = 5
        rng = np.random.default_rng(seed)
        L = 900 if m * n >= 1 << 20 else 150
        sxy = np.stack([rng.uniform(0, n - 1, L), rng.uniform(0, m - 1, L)], 1)
        suv = rng.normal(0, 2, (L, 2))
        for k in (20, 5):
            r["idw k%d" % k] = digest(idw_to_device(sxy, suv, m, n, k=k).to_host())
        res[key] = r
        del fd, preps, eig
        synchronize()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/lk_bitcheck_%s.json" % tag, "w") as f:
        json.dump(res, f, indent=1)
    print("lk_bitcheck %s: %d cases x %d fingerprints" % (tag, len(res), len(next(iter(res.values())))))


def diff(a, b):
    ra = json.load(open("gpurun_out/lk_bitcheck_%s.json" % a))
    rb = json.load(open("gpurun_out/lk_bitcheck_%s.json" % b))
    bad = [(c, k) for c in ra for k in ra[c] if ra[c][k] != rb.get(c, {}).get(k)]
    total = sum(len(v) for v in ra.values())
    print("%d of %d fingerprints differ" % (len(bad), total))
    for c, k in bad:
        print("  ", c, "|", k, ra[c][k], rb.get(c, {}).get(k))
    return 1 if bad else 0


if __name__ == "__main__":
    if sys.argv[1] == "--diff":
        sys.exit(diff(sys.argv[2], sys.argv[3]))
    run(sys.argv[1])
