#!/bin/bash
TAG=${1:-r1}
export TMPDIR=/tmp
OUT=gpurun_out/prof_structured_$TAG
mkdir -p $OUT
CMD="python3 tools/perf_structured.py 18 64 10 c64"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
grep -h "propagate\|energy_grad\|passes" $OUT/trace.log
