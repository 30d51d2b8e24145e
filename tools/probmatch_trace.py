"""Per-call kernel breakdown of a rocprofv3 kernel trace of tools/probmatch_quick.py (development aid).

    python tools/probmatch_trace.py gpurun_out/pm/prof/pm_kernel_trace.csv [call index ...]
"""
import csv
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pm_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda s: re.search(r"(pm_\w+(<\w+>)?)", s).group(1)
seq = [(name(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
calls, cur = [], []
for k in seq:  # a call starts with pm_init
    if k[0] == "pm_init" and cur:
        calls.append(cur)
        cur = []
    cur.append(k)
calls.append(cur)
print(len(calls), "calls")
for ci in [int(a) for a in sys.argv[2:]] or range(len(calls)):
    print("call %d: %.1f us in kernels" % (ci, sum(k[1] for k in calls[ci])))
    if len(sys.argv) > 2:
        for k in calls[ci]:
            print("  %-24s %8.1f" % k)
