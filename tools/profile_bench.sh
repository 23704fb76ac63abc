#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box through gpurun). $1 = tag
TAG=${1:-r1}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python3 bench.py --steps 4 --warmup 1 --skip-cpu-baseline --skip-other-configs --skip-general-path"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -30
tail -2 $OUT/trace.log
