"""GPU idle gaps between consecutive kernels of one bench step, from a rocprofv3 kernel trace."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n)
    return n.split("(")[0].replace("psh::", "")[:26]
# find the last full step: from a lk_stats1 to the following extrapolation kernel (semilag_window; semilag_fused before round 5)
idx = [i for i, r in enumerate(rows) if (nm(r).startswith("semilag_window") or nm(r).startswith("semilag_fused<1, 1, true, 4"))]
end = idx[-1]; start = idx[-2] + 1
prev_end = int(rows[idx[-2]]["End_Timestamp"])
busy = 0; total_gap = 0
for r in rows[start:end + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    total_gap += max(gap, 0); busy += (e - s) / 1e3
    print("%-26s gap %7.1f us  dur %7.1f us" % (nm(r), gap, (e - s) / 1e3))
    prev_end = e
print("busy %.1f us, gaps %.1f us" % (busy, total_gap))
