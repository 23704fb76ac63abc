#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/prof_sq
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > $OUT/sq_counters.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc -- python3 bench.py --steps 3 --warmup 1 --skip-cpu-baseline --skip-other-configs --skip-general-path > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt | cut -c1-300
python3 - <<'PY'
import csv, glob
from collections import defaultdict
fs = glob.glob("gpurun_out/prof_sq/pmc/*/*_counter_collection.csv")
if fs:
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "zgemm" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][-28:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["dur_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    for k, d in acc.items():
        print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
