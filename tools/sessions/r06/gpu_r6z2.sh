#!/bin/bash
# round 6, session z2: PYSTEPS_HIP_SL_STATS=1 - wave-passes through the window / through the gathers and window fills of the final
# kernel on the sheared, uniform and calm test fields (every launch then waits and prints: the timings of such runs do not count)
for f in sheared uniform calm; do PYSTEPS_HIP_SL_STATS=1 timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | grep -v "^{" | tail -4; done
