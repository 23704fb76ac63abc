#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/prof_structured_groups
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/perf_structured.py 18 64 10 c64 > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_structured_groups/trace/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "sapply" in r["Name"] or "soverlap" in r["Name"]:
        print(r["Name"].split("(anonymous namespace)::")[-1].split("(")[0], r["Calls"], "avg %.4f ms" % (float(r["AverageNs"]) / 1e6))
PY
