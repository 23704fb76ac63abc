"""Condense rocprofv3 CSV output (tools/profile_bench.sh) into a small text summary for profiles/.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB,
collected in separate --pmc passes, and on gfx950 FETCH_SIZE reports half of the bytes of a wide (16 B/lane)
coalesced streaming read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-B stores.
"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
out = []


def short(name):
    name = name.replace("void ", "").replace("dq::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:60]


stats = glob.glob(root + "/trace/*/*_kernel_stats.csv")
if stats:
    import os
    cmd = open(root + "/cmd.txt").read().strip() if os.path.exists(root + "/cmd.txt") else "python3 bench.py --steps 4 --warmup 1 --skip-cpu-baseline --skip-other-configs --skip-general-path"
    out.append("== rocprofv3 --kernel-trace --stats ({}) ==".format(cmd))
    out.append("%-62s %6s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for r in csv.DictReader(open(stats[0])):
        if float(r["Percentage"]) < 0.05:
            continue
        out.append("%-62s %6s %12.3f %12.4f %7s" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                   float(r["AverageNs"]) / 1e6, r["Percentage"]))

per = {}
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob(root + "/" + tag + "/*/*_counter_collection.csv")
    if not files:
        continue
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != ctr:
            continue
        k = short(r["Kernel_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    per[ctr] = acc
if per:
    out.append("")
    out.append("== HBM traffic per launch from PMC (separate passes; FETCH x2 correction for gfx950) ==")
    out.append("%-62s %6s %14s %14s %14s" % ("kernel", "calls", "read_MB", "write_MB", "total_MB"))
    keys = sorted(set(per.get("FETCH_SIZE", {})) | set(per.get("WRITE_SIZE", {})))
    for k in keys:
        f = per.get("FETCH_SIZE", {}).get(k, [0, 0.0])
        w = per.get("WRITE_SIZE", {}).get(k, [0, 0.0])
        n = max(f[0], w[0], 1)
        rd = 2.0 * f[1] * 1024 / max(f[0], 1) / 1e6
        wr = w[1] * 1024 / max(w[0], 1) / 1e6
        if rd + wr < 1.0:
            continue
        out.append("%-62s %6d %14.1f %14.1f %14.1f" % (k, n, rd, wr, rd + wr))
print("\n".join(out))
