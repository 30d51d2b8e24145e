"""Generate tests/golden/*.npz from the REAL reference (run in the build container only).

    python -m tools.make_golden

Each fixture stores seeded inputs, the keyword arguments and the outputs of the
unmodified reference function, so the GPU box (which has no /root/reference)
can check both the oracle and the HIP path against the reference's own results.
"""

import os
import sys

import numpy as np

from . import ref_loader, synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def semilag_cases():
    m, n = 72, 96
    P = synth.rain_field_db(m, n, seed=7, sigma=2.0)
    V = synth.true_velocity(m, n)
    y, x = np.mgrid[0:m, 0:n]
    Vs = V + np.stack([0.04 * (x - n / 2), -0.03 * (y - m / 2)]).astype(np.float32)  # shear, leaves domain
    Pn = P.copy()
    Pn[synth.border_nan_mask(m, n, 0.15)] = np.nan
    rng = np.random.default_rng(3)
    D0 = rng.normal(0, 3, (2, m, n))
    # non-finite velocities (allow_nonfinite_values=True, reference :106-137): a NaN block, an
    # isolated NaN, one +inf and one -inf component
    Vn = Vs.copy()
    Vn[:, 20:30, 40:55] = np.nan
    Vn[0, 50, 10] = np.nan
    Vn[1, 5, 80] = np.inf
    Vn[0, 60, 70] = -np.inf
    return {
        "sl_velnan": dict(precip=P, velocity=Vn, timesteps=4, kw=dict(allow_nonfinite_values=True, outval=-15.0)),
        "sl_velnan_K3_nanfield": dict(precip=Pn, velocity=Vn, timesteps=[0.5, 1.5, 3.0],
                                      kw=dict(allow_nonfinite_values=True, n_iter=3)),
        "sl_velnan_K0_o0": dict(precip=P, velocity=Vn, timesteps=3,
                                kw=dict(allow_nonfinite_values=True, n_iter=0, interp_order=0, outval=-15.0)),
        "sl_velnan_min": dict(precip=P, velocity=Vn, timesteps=2, kw=dict(allow_nonfinite_values=True, outval="min")),
        "sl_int_T6": dict(precip=P, velocity=V, timesteps=6, kw={}),
        "sl_shear_K3": dict(precip=P, velocity=Vs, timesteps=4, kw=dict(n_iter=3)),
        "sl_K0": dict(precip=P, velocity=Vs, timesteps=3, kw=dict(n_iter=0)),
        "sl_list_vt": dict(precip=P, velocity=Vs, timesteps=[0.5, 1.0, 2.5, 3.0], kw=dict(vel_timestep=2.0, n_iter=2)),
        "sl_nan_min": dict(precip=Pn, velocity=V, timesteps=3, kw=dict(outval="min", allow_nonfinite_values=True)),
        "sl_nan_nan": dict(precip=Pn, velocity=Vs, timesteps=3, kw=dict(allow_nonfinite_values=True)),
        "sl_order0": dict(precip=P, velocity=Vs, timesteps=3, kw=dict(interp_order=0, outval=-15.0)),
        "sl_resume": dict(precip=P, velocity=Vs, timesteps=[1.5], kw=dict(displacement_prev=D0, n_iter=1)),
        "sl_resume_K0": dict(precip=P, velocity=Vs, timesteps=[0.5, 1.5], kw=dict(displacement_prev=D0, n_iter=0)),
        "sl_order3": dict(precip=P, velocity=Vs, timesteps=3, kw=dict(interp_order=3, outval=-15.0)),
        "sl_order3_nan": dict(precip=Pn, velocity=V, timesteps=2, kw=dict(interp_order=3, allow_nonfinite_values=True)),
        "sl_f64": dict(precip=P.astype(np.float64), velocity=Vs.astype(np.float64), timesteps=2, kw={}),
        # map_coordinates_mode variants (reference :91-96, :225-232); long lead times wrap the
        # 72 x 96 domain more than once
        "sl_mode_nearest": dict(precip=P, velocity=Vs, timesteps=[4.0, 10.0], kw=dict(map_coordinates_mode="nearest")),
        "sl_mode_reflect_nan": dict(precip=Pn, velocity=Vs, timesteps=[4.0, 10.0, 40.0],
                                    kw=dict(map_coordinates_mode="reflect", allow_nonfinite_values=True)),
        "sl_mode_mirror": dict(precip=P, velocity=Vs, timesteps=[4.0, 10.0, 40.0], kw=dict(map_coordinates_mode="mirror")),
        "sl_mode_wrap": dict(precip=P, velocity=Vs, timesteps=[4.0, 10.0, 40.0], kw=dict(map_coordinates_mode="wrap")),
        "sl_mode_gridwrap": dict(precip=P, velocity=Vs, timesteps=[4.0, 10.0, 40.0],
                                 kw=dict(map_coordinates_mode="grid-wrap", n_iter=2)),
        "sl_mode_gridconst": dict(precip=P, velocity=Vs, timesteps=[4.0, 10.0],
                                  kw=dict(map_coordinates_mode="grid-constant", outval=-15.0)),
        "sl_mode_reflect_o0": dict(precip=P, velocity=Vs, timesteps=[4.0, 10.0],
                                   kw=dict(map_coordinates_mode="reflect", interp_order=0)),
        "sl_mode_gridwrap_o0": dict(precip=P, velocity=Vs, timesteps=[4.0, 10.0],
                                    kw=dict(map_coordinates_mode="grid-wrap", interp_order=0)),
    }


def make_semilag():
    ref = ref_loader.load("pysteps.extrapolation.semilagrangian")
    blob = {}
    for name, c in semilag_cases().items():
        kw = dict(c["kw"])
        out, disp = ref.extrapolate(c["precip"], c["velocity"], c["timesteps"], return_displacement=True, **kw)
        blob[name + "/precip"] = c["precip"]
        blob[name + "/velocity"] = c["velocity"]
        blob[name + "/timesteps"] = np.asarray(c["timesteps"])
        blob[name + "/timesteps_is_int"] = np.asarray(isinstance(c["timesteps"], int))
        for k, v in kw.items():
            blob[name + "/kw/" + k] = np.asarray(v)
        blob[name + "/out"] = out
        blob[name + "/disp"] = disp
    # displacement-only call (precip None), reference nowcasts/utils.py:498-503
    c = semilag_cases()["sl_shear_K3"]
    _, disp = ref.extrapolate(None, c["velocity"], [0.7], return_displacement=True, n_iter=1)
    blob["sl_disp_only/velocity"] = c["velocity"]
    blob["sl_disp_only/disp"] = disp
    np.savez_compressed(os.path.join(OUT, "semilag_reference.npz"), **blob)
    print("semilag:", len(semilag_cases()) + 1, "cases")


def semilag_order3_mode_cases():
    """interp_order=3 with the six map_coordinates modes other than "constant" (reference :91-96, :146-157,
    :225-253: prefilter with the mode's boundary kind, padded for "nearest" / "grid-constant", mask warps
    with the same mode); long lead times leave / wrap the 72 x 96 domain."""
    base = semilag_cases()
    P, Vs = base["sl_order3"]["precip"], base["sl_order3"]["velocity"]
    Pn = base["sl_order3_nan"]["precip"]
    cases = {}
    for mode in ("nearest", "reflect", "mirror", "wrap", "grid-wrap", "grid-constant"):
        tag = mode.replace("-", "")
        cases["sl_o3_" + tag] = dict(precip=P, velocity=Vs, timesteps=[2.0, 10.0, 40.0],
                                     kw=dict(interp_order=3, map_coordinates_mode=mode, outval=-15.0))
        cases["sl_o3_" + tag + "_nan"] = dict(precip=Pn, velocity=Vs, timesteps=[3.0, 12.0],
                                              kw=dict(interp_order=3, map_coordinates_mode=mode, allow_nonfinite_values=True,
                                                      n_iter=2))
    # cval = NaN padded around the field: the filter's recursions carry it into every coefficient
    cases["sl_o3_gridconstant_nancval"] = dict(precip=P, velocity=Vs, timesteps=[2.0, 10.0],
                                               kw=dict(interp_order=3, map_coordinates_mode="grid-constant"))
    return cases


def make_semilag_order3_modes():
    ref = ref_loader.load("pysteps.extrapolation.semilagrangian")
    blob = {}
    for name, c in semilag_order3_mode_cases().items():
        kw = dict(c["kw"])
        out, disp = ref.extrapolate(c["precip"], c["velocity"], c["timesteps"], return_displacement=True, **kw)
        blob[name + "/precip"] = c["precip"]
        blob[name + "/velocity"] = c["velocity"]
        blob[name + "/timesteps"] = np.asarray(c["timesteps"])
        blob[name + "/timesteps_is_int"] = np.asarray(False)
        for k, v in kw.items():
            blob[name + "/kw/" + k] = np.asarray(v)
        blob[name + "/out"] = out
        blob[name + "/disp"] = disp
    np.savez_compressed(os.path.join(OUT, "semilag_order3_modes.npz"), **blob)
    print("semilag order 3 x modes:", len(semilag_order3_mode_cases()), "cases")


def semilag_spline_order_cases():
    """interp_order 2, 4 and 5 (reference :85-90, :146-157, :225-253: B-spline prefilter of that order + the two
    order-1 mask warps), mode "constant" with and without missing values, and one other boundary mode each."""
    base = semilag_cases()
    P, Vs, Pn, V = base["sl_order3"]["precip"], base["sl_order3"]["velocity"], base["sl_order3_nan"]["precip"], base["sl_order3_nan"]["velocity"]
    cases = {}
    for order, mode in ((2, "reflect"), (4, "nearest"), (5, "grid-wrap")):
        cases["sl_o%d" % order] = dict(precip=P, velocity=Vs, timesteps=3, kw=dict(interp_order=order, outval=-15.0))
        cases["sl_o%d_nan" % order] = dict(precip=Pn, velocity=V, timesteps=2, kw=dict(interp_order=order, allow_nonfinite_values=True))
        cases["sl_o%d_%s" % (order, mode.replace("-", ""))] = dict(
            precip=P, velocity=Vs, timesteps=[2.0, 10.0, 40.0], kw=dict(interp_order=order, map_coordinates_mode=mode, outval=-15.0))
    cases["sl_o4_gridconstant_nan"] = dict(precip=Pn, velocity=Vs, timesteps=[3.0, 12.0],
                                           kw=dict(interp_order=4, map_coordinates_mode="grid-constant", allow_nonfinite_values=True, n_iter=2))
    return cases


def make_semilag_spline_orders():
    ref = ref_loader.load("pysteps.extrapolation.semilagrangian")
    blob = {}
    for name, c in semilag_spline_order_cases().items():
        kw = dict(c["kw"])
        out, disp = ref.extrapolate(c["precip"], c["velocity"], c["timesteps"], return_displacement=True, **kw)
        blob[name + "/precip"] = c["precip"]
        blob[name + "/velocity"] = c["velocity"]
        blob[name + "/timesteps"] = np.asarray(c["timesteps"])
        blob[name + "/timesteps_is_int"] = np.asarray(isinstance(c["timesteps"], int))
        for k, v in kw.items():
            blob[name + "/kw/" + k] = np.asarray(v)
        blob[name + "/out"] = out
        blob[name + "/disp"] = disp
    np.savez_compressed(os.path.join(OUT, "semilag_spline_orders.npz"), **blob)
    print("semilag spline orders 2 / 4 / 5:", len(semilag_spline_order_cases()), "cases")


def semilag_xy_cases():
    """Custom ``xy_coords`` (reference :68-72, :174-179: the positions the trajectories start from - callers pass a
    sub-grid or, as here, a deformed grid): a smooth sub-pixel deformation, the same with a displacement_prev, a
    staggered half-pixel grid without the midpoint rule, interp_order 0 and a displacement-only call."""
    base = semilag_cases()
    P, Vs, Pn = base["sl_shear_K3"]["precip"], base["sl_shear_K3"]["velocity"], base["sl_nan_nan"]["precip"]
    m, n = P.shape
    y, x = np.mgrid[0:m, 0:n].astype(np.float64)
    warp = np.stack([x + 1.7 * np.sin(y / 9.0) + 0.3, y + 2.2 * np.cos(x / 11.0) - 0.6])
    half = np.stack([x + 0.5, y + 0.5])
    D0 = base["sl_resume"]["kw"]["displacement_prev"]
    return {
        "sl_xy_warp": dict(precip=P, velocity=Vs, timesteps=3, kw=dict(xy_coords=warp, outval=-15.0)),
        "sl_xy_warp_resume": dict(precip=P, velocity=Vs, timesteps=[0.5, 2.0], kw=dict(xy_coords=warp, displacement_prev=D0, n_iter=2)),
        "sl_xy_half_K0": dict(precip=Pn, velocity=Vs, timesteps=[1.0, 2.5], kw=dict(xy_coords=half, n_iter=0, allow_nonfinite_values=True)),
        "sl_xy_warp_o0": dict(precip=P, velocity=Vs, timesteps=2, kw=dict(xy_coords=warp, interp_order=0, outval=-15.0)),
        "sl_xy_warp_o3": dict(precip=P, velocity=Vs, timesteps=2, kw=dict(xy_coords=warp, interp_order=3, outval=-15.0)),
    }


def make_semilag_xy():
    ref = ref_loader.load("pysteps.extrapolation.semilagrangian")
    blob = {}
    for name, c in semilag_xy_cases().items():
        kw = dict(c["kw"])
        out, disp = ref.extrapolate(c["precip"], c["velocity"], c["timesteps"], return_displacement=True, **kw)
        blob[name + "/precip"] = c["precip"]
        blob[name + "/velocity"] = c["velocity"]
        blob[name + "/timesteps"] = np.asarray(c["timesteps"])
        blob[name + "/timesteps_is_int"] = np.asarray(isinstance(c["timesteps"], int))
        for k, v in kw.items():
            blob[name + "/kw/" + k] = np.asarray(v)
        blob[name + "/out"] = out
        blob[name + "/disp"] = disp
    c = semilag_xy_cases()["sl_xy_warp"]
    _, disp = ref.extrapolate(None, c["velocity"], [0.7, 1.9], return_displacement=True, n_iter=1, xy_coords=c["kw"]["xy_coords"])
    blob["sl_xy_disp_only/velocity"] = c["velocity"]
    blob["sl_xy_disp_only/xy_coords"] = c["kw"]["xy_coords"]
    blob["sl_xy_disp_only/disp"] = disp
    np.savez_compressed(os.path.join(OUT, "semilag_xy_coords.npz"), **blob)
    print("semilag xy_coords:", len(semilag_xy_cases()) + 1, "cases")


def sparse_vectors(L, m, n, seed, outliers=True):
    """LK-like sparse vectors: integer feature positions, smooth motion + noise (+ planted outliers)."""
    rng = np.random.default_rng(seed)
    xy = np.column_stack([rng.integers(0, n, L), rng.integers(0, m, L)]).astype(float)
    uv = np.column_stack([
        4 + 2 * np.sin(2 * np.pi * xy[:, 1] / m) + rng.normal(0, 0.15, L),
        -3 + 1.5 * np.cos(2 * np.pi * xy[:, 0] / n) + rng.normal(0, 0.15, L),
    ])
    if outliers:
        bad = rng.choice(L, max(1, L // 40), replace=False)
        uv[bad] += rng.choice([-1, 1], (bad.size, 2)) * rng.uniform(4, 9, (bad.size, 2))
    return xy, uv.astype(np.float32).astype(float)


def make_sparse():
    cl = ref_loader.load("pysteps.utils.cleansing")
    ip = ref_loader.load("pysteps.utils.interpolate")
    blob = {}
    for name, (L, m, n, seed) in {"a": (400, 120, 160, 1), "b": (60, 90, 70, 2), "c": (1500, 256, 256, 3)}.items():
        xy, uv = sparse_vectors(L, m, n, seed)
        out = cl.detect_outliers(uv, 3, xy, 30)
        dxy, duv = cl.decluster(xy[~out], uv[~out], 20, 1)
        blob["%s/xy" % name], blob["%s/uv" % name] = xy, uv
        blob["%s/shape" % name] = np.array([m, n])
        blob["%s/outliers" % name] = out
        blob["%s/dxy" % name], blob["%s/duv" % name] = dxy, duv
        if name != "c":
            blob["%s/idw" % name] = ip.idwinterp2d(dxy, duv, np.arange(n), np.arange(m))
            blob["%s/idw_k5_p2" % name] = ip.idwinterp2d(dxy, duv, np.arange(n), np.arange(m), power=2.0, k=5, dist_offset=0.1)
    np.savez_compressed(os.path.join(OUT, "sparse_reference.npz"), **blob)
    print("sparse: 3 cases")


def probmatch_cases():
    """(initial, target) pairs as the member loops produce them: a continuous forecast with its dry
    pixels at the zero value against a quantised observation; wet values of the initial arrays are
    tie-free (the reference's unstable argsort leaves tied wet values unspecified)."""
    rng = np.random.default_rng(11)
    m, n = 96, 128
    obs = np.round(synth.rain_field_db(m, n, seed=5, sigma=3.0).astype(float), 1)
    fct = synth.rain_field_db(m, n, seed=6, sigma=3.0).astype(float) + rng.normal(0, 1e-3, (m, n))
    cases = {}
    dry = fct.copy()
    dry[dry < np.percentile(dry, 70)] = -15.0
    wetobs = obs.copy()
    wetobs[wetobs < np.percentile(wetobs, 40)] = -15.0
    dryobs = obs.copy()
    dryobs[dryobs < np.percentile(dryobs, 90)] = -15.0
    nanobs = wetobs.copy()
    nanobs[synth.border_nan_mask(m, n, 0.1)] = np.nan
    cases["adjust"] = (dry, wetobs)          # more rain in the target: percentile threshold (:107-110)
    cases["no_adjust"] = (dry, dryobs)       # less rain in the target
    cases["nan_target"] = (dry, nanobs)      # NaNs of the target count as zeros (:103-104)
    cases["all_wet"] = (fct, wetobs)         # no mask applied: every pixel above the single minimum
    cases["gauss"] = (rng.normal(size=(64, 64)), rng.normal(size=(64, 64)))
    return cases


def make_probmatch():
    pm = ref_loader.load("pysteps.postprocessing.probmatching")
    blob = {}
    for name, (initial, target) in probmatch_cases().items():
        blob[name + "/initial"], blob[name + "/target"] = initial, target
        blob[name + "/out"] = pm.nonparam_match_empirical_cdf(initial, target)
    np.savez_compressed(os.path.join(OUT, "probmatch_reference.npz"), **blob)
    print("probmatch: %d cases" % (len(blob) // 3))


def main():
    if not ref_loader.available():
        sys.exit("reference not available")
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    for name, fn in (("semilag", make_semilag), ("semilag_order3_modes", make_semilag_order3_modes),
                     ("semilag_xy", make_semilag_xy), ("semilag_spline_orders", make_semilag_spline_orders), ("sparse", make_sparse), ("probmatch", make_probmatch)):
        if not only or name in only:
            fn()


if __name__ == "__main__":
    main()
