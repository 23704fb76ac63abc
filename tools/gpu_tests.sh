#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/gpu_tests.sh [pytest args]
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu "$@" 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
