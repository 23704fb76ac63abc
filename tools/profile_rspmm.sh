#!/bin/bash
# PMC passes over the sparse-pass probe (tools/_build/rspmm_probe built with -DPROBE_ONLY_OLD: library kernels only):
# L2 hit / miss / requests, L1 -> L2 read requests and their latency, fabric reads, per kernel.
export TMPDIR=/tmp
OUT=gpurun_out/prof_rspmm
mkdir -p $OUT
BIN=tools/_build/rspmm_probe_lib
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/p1 -- $BIN 12 10 > $OUT/p1.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/p2 -- $BIN 12 10 > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p3 -- $BIN 12 10 > $OUT/p3.log 2>&1
rocprofv3 --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace --output-format csv -d $OUT/p4 -- $BIN 12 10 > $OUT/p4.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/prof_rspmm/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void dq::", "")[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-34s %14.1f" % (c, sum(v) / len(v)))
PY
