#!/bin/bash
# rocprofv3 passes over the headline bench command (state-column form): kernel trace + stats, HBM-side traffic (separate
# FETCH_SIZE / WRITE_SIZE passes; FETCH x2 on gfx950, guide's HBM section) and SQ counter passes that say where the waves of
# xs_forward_kernel / xs_backward_kernel spend their time.   tools/profile_statecol.sh TAG [extra bench.py args...]
# The program after `--` is python3 itself (no env / bash -c hop); counters never share a run with sys/hip traces.
TAG=${1:-r3_statecol}; shift 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python3 bench.py --steps 20 --warmup 2 --trials 1 --skip-cpu-baseline --skip-other-configs --skip-general-path --extras $OUT/extras.json $*"
echo "$CMD" > $OUT/cmd.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/sq1 -- $CMD > $OUT/sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.log 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/tcc -- $CMD > $OUT/tcc.log 2>&1 || echo "tcc pass failed" >> $OUT/trace.log
python3 tools/summarize_prof.py $OUT > gpurun_out/${TAG}_summary.txt
python3 - $OUT >> gpurun_out/${TAG}_summary.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/sq*/*/*_counter_collection.csv") + glob.glob(root + "/tcc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void dq::", "")[:44]
        if "xs_" not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("\n== SQ / TCC counters per launch (averages; SQ cycle counters in quad-cycles, summed over the chip) ==")
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.1f   (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
tail -2 $OUT/trace.log
