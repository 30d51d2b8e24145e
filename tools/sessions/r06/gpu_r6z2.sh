for f in sheared uniform calm; do PYSTEPS_HIP_SL_STATS=1 timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | grep -v "^{" | tail -4; done
