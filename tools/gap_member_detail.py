"""Kernels and idle gaps of ONE member update of the resident STEPS loop in launch order, from a rocprofv3 kernel trace
of bench.py's steps-loop leg (tools/gpu_member_round.sh): everything on the main stream between two consecutive
launches of the marker kernel (default fft_rows_r2c: the first kernel of a spectral member update).
    python tools/gap_member_detail.py <trace dir> [marker]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
marker = sys.argv[2] if len(sys.argv) > 2 else "fft_rows_r2c"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n)
    return n.split("(")[0].replace("psh::", "")[:34]
rows = [r for r in rows if not nm(r).startswith(("mt_produce", "polar_"))]  # the generators' side stream
idx = [i for i, r in enumerate(rows) if nm(r).startswith(marker)]
start, end = idx[-4], idx[-3]
prev_end = int(rows[start - 1]["End_Timestamp"])
busy = gaps = 0.0
for r in rows[start:end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    gaps += max(gap, 0); busy += (e - s) / 1e3
    print("%-34s gap %7.1f us  dur %7.1f us" % (nm(r), gap, (e - s) / 1e3))
    prev_end = max(prev_end, e)
print("busy %.1f us, gaps %.1f us" % (busy, gaps))
