"""HBM bytes of ONE STEPS member update from a tools/gpu_member_round.sh session: per kernel of the update (the kernels
between two noise draws in member_update_kernels.txt, calls / members per update) the FETCH_SIZE / WRITE_SIZE means of
the separate --pmc passes (member_pmc.csv; FETCH_SIZE doubled: gfx950 reports half of a coalesced read stream,
tools/calib_copy.py) -> profiles/<round>/<prefix>_member_update_hbm_bytes.txt and profiles/member_update_traffic.json
(with the git blob hashes of the update's kernel sources: bench.py withholds the figure for another source revision).
    python tools/member_traffic.py <gpurun_out/<tag>_member> <round> <prefix> [members]"""
import hashlib, json, os, re, sys

src, rnd, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
members = int(sys.argv[4]) if len(sys.argv) > 4 else 6
MU_SOURCES = ("pysteps_amd/csrc/steps_loop.hip", "pysteps_amd/csrc/fft.hip", "pysteps_amd/csrc/probmatch.hip",
              "pysteps_amd/csrc/mask.hip", "pysteps_amd/csrc/cascade.hip", "pysteps_amd/csrc/common.h")
SKIP = ("polar_", "mt_produce", "convert_elements", "semilag_members", "__amd_rocclr")


def blob_hash(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


calls = {}
for line in open(os.path.join(src, "member_update_kernels.txt")):
    m = re.match(r"(\S.*?)\s+calls\s+(\d+)\s+total\s+([\d.]+) us", line)
    if m and not m.group(1).startswith(SKIP):
        calls[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
pmc = {}
for line in list(open(os.path.join(src, "member_pmc.csv")))[1:]:
    # kernel names carry commas (template arguments): the three last fields are counter, value, launches
    kernel, counter, value, _ = line.rstrip("\n").rsplit(",", 3)
    pmc.setdefault(kernel.strip(), {})[counter] = float(value)
dst = os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
total, lines = 0.0, []
for name, (n, us) in calls.items():
    c = pmc.get(name)
    if c is None:  # template arguments / names cut at 48 characters
        c = next((v for k, v in pmc.items() if (k.startswith(name) and k[len(name):len(name) + 1] in "<(") or
                  (len(k) >= 48 and name.startswith(k))), None)
    if c is None or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        lines.append("%-28s (no counters)" % name)
        continue
    per_update = n / float(members)
    fetch_mb, write_mb = c["FETCH_SIZE"] / 1024.0 * per_update, c["WRITE_SIZE"] / 1024.0 * per_update
    mb = 2.0 * fetch_mb + write_mb
    total += mb
    lines.append("%-28s fetch %8.1f MB x2  write %8.1f MB -> %8.1f MB   %7.1f us per update" % (name, fetch_mb, write_mb, mb, us / members))
lines.append("HBM bytes per member update (FETCH_SIZE doubled as calibrated in round 1): %.2f GB" % (total / 1024.0))
open(os.path.join(dst, "%s_member_update_hbm_bytes.txt" % prefix), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
json.dump({"hbm_bytes_per_member_update": total * 1024.0 * 1024.0,
           "workload": "4096x4096, 6 cascade levels, AR(2), incremental mask, CDF matching",
           "source": "profiles/%s/%s_member_update_hbm_bytes.txt (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, "
                     "tools/gpu_member_round.sh + tools/member_traffic.py; FETCH_SIZE doubled as calibrated in round 1)" % (rnd, prefix),
           "source_hashes": {s: blob_hash(s) for s in MU_SOURCES}},
          open(os.path.join("profiles", "member_update_traffic.json"), "w"), indent=1)
