#!/usr/bin/env python3
"""Benchmark of the propagator hot path on MI355X.  Default = the headline: Trotter steps/s (forward + adjoint gradient),
n = 12 qubits, batch = 4096 per GPU (BASELINE.json configs[2]; configs[3] = the same per-GPU load on 8 GPUs).

    python bench.py --gpus N --steps K --warmup W            # N > 1: starts N child ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               # ... or is started as one rank of N by a launcher

Rank 0 prints ONE JSON line.  `--config 3` (default) is the headline; `--config 2` is BASELINE configs[1] (8-qubit ring
QAOA, 1000 steps, batch 1024, forward); `--config 5` is configs[4] (18 qubits, complex64, batch 512 over the GPUs, the
matrix-free engine, HBM-bound).  `--workload dense-real | dense-complex` replaces the Ising + X_k inputs of config 3 by
dense random matrices of the same size and step norm (what `set_H`, diffqc.cc:43-67, accepts in general).

A "step" (config 3) is ONE Trotter step of the fwd+adjoint pass over one GPU's whole batch: 12-qubit random Ising H stored
dense (d = 4096), K = 12 controls X_k, B = 4096 states, fp64 complex.  The timed region is one `dq_energy_grad` call over a
chain of K steps -- K forward steps (assembly, exp, U.Psi), then the K-step adjoint sweep -- plus, for N > 1, the
collectives of the multi-GPU form.  At this shape (batch >= dim) the backward sweep runs in operator form and keeps no
state checkpoints, so the per-step cost does not depend on the chain length (`other_configs.full_chain_S2000` is the whole
2000-step configuration in one call).  Inputs (H0, H_k, M, Psi0, pulse table) are resident in HBM before the clock starts.
N GPUs = N batch shards of 4096 (weak scaling): value = N*K / max-over-ranks wall time.

THE HEADLINE NUMBER DEPENDS ON THE STRUCTURE OF THE INPUTS.  The engine inspects H0 / H_k at `dq_set_hamiltonians`: real
matrices take the cos X + i sin X form on real products, sparse controls their nonzeros only, a sparse drift too turns the
whole step exponential into sparse x dense passes, and a generator that is a sum of Pauli strings (XOR-regular pattern)
builds every column of U in LDS with no dense intermediate at all.  BASELINE's workload (diagonal Ising drift, X_k controls)
is of the last kind.  `by_input_structure` in the JSON line gives, in the same run and with their own `roofline` blocks, the
rates for dense real symmetric and dense complex Hermitian inputs of the same size (random data, 20-step chains), and for
the Ising inputs with the structure ignored step by step (gradients equal to rounding).

JSON extras: `roofline` = the kernel class with the largest share of the timed region, decided live from the per-launch
HIP events the library records on the launch stream.  For MFMA classes `achieved` is the EXECUTED matrix-pipe rate (the
3M complex product executes 6*M*N*K flops for 8*M*N*K algorithmic ones; the Hermitian-output form only the tiles on and
above the diagonal), so `frac` = achieved / peak is a fraction of the matrix pipe; the algorithmic rate is
`achieved_algorithmic`.  `traffic` = HBM-side bytes per launch from rocprofv3 PMC passes (`traffic_source` names the
profiles/ file), `traffic_over_algorithmic` its ratio to the algorithmic bytes.  `cpu_baseline` = the oracle (the
reference's scipy expm + matmul step) on this box's host cores, plus the single-thread C restatement of diffqc.cc.
"""
import argparse
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# measuring, reporting and launching live in benchlib/ (report.py: roofline blocks + the ONE line; launch.py: ranks; measure.py: the
# timed regions).  The names the contract tests use stay importable from here.
from benchlib import report  # noqa: E402
from benchlib.launch import Dist, launch_ranks  # noqa: E402,F401
from benchlib.measure import run_config1, run_config2, run_config3, run_config5  # noqa: E402
from benchlib.report import (LINE_LIMIT, PEAK_F64_MFMA_TFLOPS, PEAK_HBM_GBPS, PEAK_LDS_GBPS, TRAFFIC, claim_stdout, compact_line,  # noqa: E402,F401
                             executed_tflops, ranked, roofline_blocks, short_roofline)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed Trotter steps (default 20; config 2: 1000)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=(1, 2, 3, 5),
                    help="3 = headline (BASELINE configs[2]/[3]); 1 = configs[0] (demo_maxcut.py: training epochs/s, oracle CPU loop beside it); "
                         "2 = configs[1] (n=8 QAOA forward); 5 = configs[4] (n=18, c64, matrix-free)")
    ap.add_argument("--workload", choices=("ising", "dense-real", "dense-complex", "xy-drives", "qaoa-ring"), default="ising",
                    help="config 3 inputs: BASELINE's random Ising + X_k (default) or dense random matrices of the same size and norm")
    ap.add_argument("--qubits", type=int, default=12)
    ap.add_argument("--batch", type=int, default=None, help="states per GPU (config 5: the WHOLE batch, default 512)")
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-other-configs", action="store_true")
    ap.add_argument("--skip-full-chain", action="store_true", help="do not run the full S = 2000 chain extra (about a minute)")
    ap.add_argument("--skip-gemm-chain-check", action="store_true",
                    help="full chain: do not repeat the 2000-step gradient on the checkpointed GEMM chain (about half a minute)")
    ap.add_argument("--skip-general-path", action="store_true",
                    help="do not run the by_input_structure extras (profiling: only the headline path's kernels)")
    ap.add_argument("--chain", type=int, default=2000,
                    help="length of the full configuration's chain: the timed slice uses ITS checkpoint stride")
    ap.add_argument("--real-path", type=int, default=-1,
                    help="-1 auto (real H -> cos/sin form on real MFMA products), 0 = general complex path only")
    ap.add_argument("--state-columns", type=int, default=-1, choices=(-1, 0, 1),
                    help="config 3: -1 auto (Pauli-sum generators run the state-column form), 0 = the explicit-propagator GEMM chain as the "
                         "timed workload (profiling the north-star kernel on the headline's inputs), 1 = whenever the kernels exist")
    ap.add_argument("--multi-gpu", choices=("auto", "sliced", "replicated", "propagators", "columns"), default="replicated",
                    help="N > 1: 'replicated' (default; north_star / SURVEY 8-E1) = every rank runs the whole chain on its batch shard, one "
                         "all-reduce of loss + dL/du; 'sliced' = chain split over time across the ranks (operator form; three collectives); "
                         "auto = both checked and tried, the faster one timed; 'propagators' = SURVEY 8-F2 as written: state-form "
                         "chain on propagators built 1/N per rank and all-gathered (never chosen by auto: 0.8 GB of collectives per step)")
    ap.add_argument("--no-beside", action="store_true",
                    help="N > 1: do not measure, beside the timed form, the sliced form and the whole global batch on one GPU")
    ap.add_argument("--trials", type=int, default=3, help="how many times the K-step region is timed (the line reports the median)")
    ap.add_argument("--extras", default=None, help="where the full record goes (default: bench_extras.json beside bench.py)")
    ap.add_argument("--launch", action="store_true", help="start the ranks through torch.distributed.run even for --gpus 1")
    a = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.launch):
        sys.exit(launch_ranks(a, sys.argv[1:]))
    claim_stdout()
    if a.extras:
        report.set_extras_path(os.path.abspath(a.extras))
    D = Dist(a)
    {1: run_config1, 2: run_config2, 3: run_config3, 5: run_config5}[a.config](a, D)
    D.close()  # (only on success: a failing rank must not enter a barrier its peers will never reach -- it exits non-zero)


if __name__ == "__main__":
    main()

