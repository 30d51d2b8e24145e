#!/bin/bash
# round 6, session B: the window kernel on rows that are not 16-byte aligned (n % 4 != 0): fingerprints with the window forced
# against the gather kernels, the extrapolator's suites, and what a 640 x 710 / 1226 x 761 call costs either way
OUT=gpurun_out/${1:-r6B}; mkdir -p $OUT
export TMPDIR=/tmp
{
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py w12 2>&1 | tail -1
python tools/sl_bitcheck.py --diff v7 w12 | tail -1
timeout 300 python tools/sl_bitcheck.py auto 2>&1 | tail -1
python tools/sl_bitcheck.py --diff v7 auto | tail -1
timeout 900 python -m pytest tests/test_semilag_gpu.py tests/test_callers_gpu.py tests/test_nowcast_gpu.py -q -m gpu -x --timeout=400 2>&1 | tail -3
PYSTEPS_HIP_SL_VARIANT=12 timeout 900 python -m pytest tests/test_semilag_gpu.py -q -m gpu -x --timeout=400 -k "not config5 and not config3" 2>&1 | tail -2
timeout 300 python - <<'PY'
import numpy as np, time
from pysteps_amd import _lib
from pysteps_amd.device import DeviceArray, Event, synchronize
from pysteps_amd.extrapolation import get_method
from tools import synth
ex = get_method("semilagrangian"); lib = _lib.lib()
for (m, n) in ((640, 710), (1226, 761), (2175, 1725), (640, 712)):
    p = DeviceArray.from_host(synth.rain_field_db(m, n)); v = DeviceArray.from_host(synth.true_velocity(m, n))
    for variant in (7, 0):
        _lib.check(lib.psh_set_option(b"semilag_variant", variant))
        for _ in range(3): out = ex(p, v, 24, outval=-15.0, n_iter=1)
        synchronize(); e0, e1 = Event(), Event(); e0.record()
        for _ in range(10): out = ex(p, v, 24, outval=-15.0, n_iter=1)
        e1.record(); synchronize()
        print("%dx%d T=24 variant %d (%s): %.4f ms per call" % (m, n, variant, "gathers" if variant else "default", e0.elapsed_ms(e1) / 10))
    _lib.check(lib.psh_set_option(b"semilag_variant", 0))
PY
} > $OUT/log.txt 2>&1
cat $OUT/log.txt
