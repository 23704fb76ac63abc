#!/bin/bash
# rocprofv3 passes over the headline inputs on the explicit-propagator GEMM chain (bench.py --state-columns 0): kernel trace + stats
# and the separate FETCH_SIZE / WRITE_SIZE passes -- the record behind `roofline_mfma` (zgemm_lin_kernel).   tools/profile_gemm_chain.sh TAG
TAG=${1:-r4_bench_gemm_chain}; shift 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python3 bench.py --steps 8 --warmup 2 --trials 1 --state-columns 0 --skip-cpu-baseline --skip-other-configs --skip-general-path --extras $OUT/extras.json $*"
echo "$CMD" > $OUT/cmd.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || exit 1
python3 tools/summarize_prof.py $OUT > gpurun_out/${TAG}_summary.txt
tail -2 $OUT/trace.log
