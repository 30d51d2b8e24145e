"""Kernels and idle gaps of ONE member update of the resident STEPS loop (between two consecutive
ar_recompose launches), from a rocprofv3 kernel trace of bench.py --force-members-path."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n)
    return n.split("(")[0].replace("psh::", "")[:30]
rows = [r for r in rows if not nm(r).startswith(("mt_produce", "polar_"))]  # the side stream
idx = [i for i, r in enumerate(rows) if nm(r).startswith("ar_recompose")]
start, end = idx[-3], idx[-2]
prev_end = int(rows[start - 1]["End_Timestamp"])
busy = gaps = 0.0
for r in rows[start:end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    gaps += max(gap, 0); busy += (e - s) / 1e3
    print("%-30s gap %7.1f us  dur %7.1f us" % (nm(r), gap, (e - s) / 1e3))
    prev_end = max(prev_end, e)
print("busy %.1f us, gaps %.1f us" % (busy, gaps))
