#!/bin/bash
# builds (here, cross-compiled) or runs (on the GPU box) the ablation probe of the real GEMM kernel
mkdir -p tools/_build
if [ "$1" = "build" ]; then
  for a in ${ABL:-0 1 2 6 7 8 16}; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDQ_RGEMM_ABLATE=$a tools/rgemm_probe.hip -o tools/_build/rgemm_probe_$a || exit 1
  done
else
  for a in ${ABL:-0 1 2 6 7 8 16}; do timeout 60 tools/_build/rgemm_probe_$a; done
fi
