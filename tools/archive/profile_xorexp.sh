#!/bin/bash
# PMC passes over the column-form probe (tools/_build/xorexp_probe): where do the waves of xr_forward / xr_reverse spend their time?
export TMPDIR=/tmp
OUT=gpurun_out/prof_xorexp
mkdir -p $OUT
BIN=tools/_build/xorexp_probe
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/p1 -- $BIN 12 5 3 0 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/p2 -- $BIN 12 5 3 0 > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p3 -- $BIN 12 5 3 0 > $OUT/p3.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/prof_xorexp/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void dq::", "")[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-34s %14.1f" % (c, sum(v) / len(v)))
PY
tail -2 $OUT/p3.log
