#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/prof_clock
mkdir -p $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -- python3 bench.py --steps 4 --warmup 1 --skip-cpu-baseline --skip-other-configs --skip-general-path > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_clock/pmc/*/*_counter_collection.csv")[0]
acc = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "zgemm" not in r["Kernel_Name"]:
        continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    clk = float(r["Counter_Value"]) / 8.0 / dur / 1e9
    acc.setdefault(r["Kernel_Name"].split("(")[0][-40:], []).append((clk, dur * 1e3))
for k, v in acc.items():
    print(k, "n=%d  clock %.3f GHz (min %.3f max %.3f)  dur %.3f ms" % (len(v), sum(c for c, _ in v) / len(v), min(c for c, _ in v), max(c for c, _ in v), sum(d for _, d in v) / len(v)))
PY
