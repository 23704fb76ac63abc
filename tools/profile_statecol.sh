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
cp $OUT/extras.json $OUT/extras_trace.json   # (bench.py's own record of the traced run: the later passes overwrite extras.json)
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
# every launch of the two chain kernels in the traced run, and what bench.py's own HIP events said in that same run
print("\n== launches of the chain kernels in the kernel-trace pass (ms each; the 2-step launches are the warm-up) ==")
for f in glob.glob(root + "/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void dq::", "")
        if k.startswith("xs_forward_kernel") or k.startswith("xs_backward"):
            print("   %-64s %9.3f" % (k[:64], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
try:
    import json
    ex = json.load(open(root + "/extras_trace.json"))
    print("   bench.py in the same (traced) run: %.2f steps/s; HIP events per timed launch: %s %.3f ms, %s %.3f ms" % (
        ex["value"], ex["roofline"]["kernel"][:24], ex["roofline"]["ms_per_launch"], ex["roofline_other"]["kernel"][:20],
        ex["roofline_other"]["ms_per_launch"]))
except Exception as e:
    print("   (no extras.json of the traced run: %s)" % e)
print("\n== SQ / TCC counters per launch (averages; SQ cycle counters in quad-cycles, summed over the chip) ==")
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.1f   (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
tail -2 $OUT/trace.log
