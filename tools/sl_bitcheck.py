"""Bit-level fingerprint of the semi-Lagrangian kernels over a spread of configurations
(development aid): run before and after a kernel change, the digests must not move.

    python tools/sl_bitcheck.py <tag>      -> gpurun_out/bitcheck_<tag>.json
    python tools/sl_bitcheck.py --diff a b
"""
import hashlib, json, os, sys
import numpy as np
sys.path.insert(0, ".")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def run(tag):
    from pysteps_amd.extrapolation import get_method
    from pysteps_amd.extrapolation.ensemble import EnsembleAdvector
    from tools import synth
    ex = get_method("semilagrangian")
    res = {}
    for (m, n) in ((517, 389), (1024, 1024), (64, 200)):
        p = synth.rain_field_db(m, n)
        p[synth.border_nan_mask(m, n)] = np.nan if m == 517 else p.min()
        v = synth.true_velocity(m, n) * (3.0 if m == 64 else 1.0)
        for order in (0, 1, 3):
            for k in (0, 1, 3):
                out, d = ex(p, v, [0.5, 1.0, 2.5, 4.0], outval=-15.0, n_iter=k, interp_order=order, allow_nonfinite_values=True,
                            return_displacement=True)
                res["%dx%d o%d k%d" % (m, n, order, k)] = [digest(out), digest(d)]
        # resumed trajectories
        out1, d1 = ex(p, v, 2, n_iter=1, return_displacement=True, allow_nonfinite_values=True)
        out2, d2 = ex(p, v, 2, n_iter=1, return_displacement=True, displacement_prev=d1, allow_nonfinite_values=True)
        res["%dx%d resume" % (m, n)] = [digest(out2), digest(d2)]
        adv = EnsembleAdvector(v, 3, n_iter=1)
        members = np.stack([p, p + 1.0, p * 0.5]).astype(np.float32)
        o = adv.step(members, [1.0, 1.0])
        o = adv.step(members, [1.0])
        res["%dx%d members" % (m, n)] = [digest(o), digest(adv.displacement.to_host())]
    # long calls (the packed-plane kernels): gentle and strong shear, a hole in the motion field
    for (m, n) in ((1024, 1024), (300, 260), (96, 4096)):
        p = synth.rain_field_db(m, n)
        p[synth.border_nan_mask(m, n)] = np.nan
        for gain in (1.0, 6.0):
            v = synth.true_velocity(m, n) * gain
            if gain > 1.0:
                v[:, m // 3:m // 3 + 5, n // 2:n // 2 + 9] = np.nan
            for k in (0, 1, 2):
                for order in (0, 1):
                    out, d = ex(p, v, 10, outval=-15.0, n_iter=k, interp_order=order, allow_nonfinite_values=True,
                                return_displacement=True)
                    res["long %dx%d g%g o%d k%d" % (m, n, gain, order, k)] = [digest(out), digest(d)]
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bitcheck_%s.json" % tag, "w") as f:
        json.dump(res, f, indent=1)
    print("bitcheck %s: %d fingerprints" % (tag, len(res)))


def diff(a, b):
    ra = json.load(open("gpurun_out/bitcheck_%s.json" % a))
    rb = json.load(open("gpurun_out/bitcheck_%s.json" % b))
    bad = [k for k in ra if ra[k] != rb.get(k)]
    print("%d of %d fingerprints differ" % (len(bad), len(ra)))
    for k in bad:
        print("  ", k, ra[k], rb.get(k))
    return 1 if bad else 0


if __name__ == "__main__":
    if sys.argv[1] == "--diff":
        sys.exit(diff(sys.argv[2], sys.argv[3]))
    run(sys.argv[1])
