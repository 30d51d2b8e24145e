"""All kernels (both streams) between the last two `polar_write` launches of a traced member-loop run:
name, stream-agnostic start offset, duration - to see what an update waits for."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n)
    return n.split("(")[0].replace("psh::", "")[:30]
idx = [i for i, r in enumerate(rows) if nm(r).startswith("polar_write")]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
agg = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = nm(r)
    agg.setdefault(k, [0, 0.0, (s - t0) / 1e3])
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
for k, (c, d, first) in agg.items():
    print("%-30s calls %3d  total %9.1f us  first at %9.1f us" % (k, c, d, first))
print("span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
