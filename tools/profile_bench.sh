#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box through gpurun).
#   tools/profile_bench.sh TAG [trace|all] [extra bench.py args...]
# trace: kernel-trace + stats only; all (default): + separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (the guide's HBM
# section: separate passes, FETCH x2 on gfx950).  The program after `--` is python3 itself (no env / bash -c hop).
TAG=${1:-r2}; MODE=${2:-all}; shift 2
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python3 bench.py --steps 4 --warmup 1 --skip-cpu-baseline --skip-other-configs --skip-general-path $*"
echo "$CMD" > $OUT/cmd.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || exit 1
if [ "$MODE" = "all" ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || exit 1
fi
python3 tools/summarize_prof.py $OUT > gpurun_out/${TAG}_summary.txt
tail -2 $OUT/trace.log
