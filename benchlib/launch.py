"""bench.py, launcher half: `--gpus N` starts its N ranks itself; `Dist` = one process per GPU (RCCL, or gloo for rehearsals)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def launch_ranks(a, argv):
    """Start N fresh child ranks (one per GPU) BEFORE this process makes any GPU call and pass rank 0's JSON line
    through.  The parent never initialises HIP (torch.cuda.device_count() does not, on this image), and it starts the
    ranks as child processes -- it never re-execs itself."""
    backend = os.environ.get("DQ_DIST_BACKEND", "nccl")
    if backend == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus:
            print("bench.py: --gpus {} but this node shows {} GPU(s)".format(a.gpus, have), file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH] + [x for x in argv if x != "--launch"]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode



class Dist:
    """One process per GPU.  RCCL ("nccl") on device tensors; DQ_DIST_BACKEND=gloo stages the collectives through the host
    and shares the devices round-robin (rehearsal on a box with fewer GPUs than ranks: RCCL refuses two ranks per device)."""

    def __init__(self, a):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != a.gpus:
            if self.rank == 0:
                print("bench.py: --gpus {} but WORLD_SIZE {}".format(a.gpus, self.world), file=sys.stderr)
            sys.exit(2)
        self.backend = os.environ.get("DQ_DIST_BACKEND", "nccl")
        if self.backend != "nccl":
            local = local % max(1, torch.cuda.device_count())
        self.local = local
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        self.on = self.world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or "MASTER_PORT" in os.environ
        if self.on:  # launched by torch.distributed.run (also exercised at world size 1)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            os.environ.setdefault("RANK", str(self.rank))
            os.environ.setdefault("WORLD_SIZE", str(self.world))
            import datetime
            tmo = datetime.timedelta(minutes=5)  # a mismatched collective fails the run instead of hanging it
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev, timeout=tmo)
            else:
                dist.init_process_group(self.backend, timeout=tmo)
            self.world = dist.get_world_size()

    def all_reduce_(self, t, op):  # in place on a device tensor, whatever the backend
        if not self.on:
            return
        if self.backend == "nccl":
            self.dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            self.dist.all_reduce(h, op=op)
            t.copy_(h)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.on:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if not self.on:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev)
        self.all_reduce_(t, self.dist.ReduceOp.MAX)
        return float(t.item())

    def name(self):
        return ("nccl (RCCL over xGMI)" if self.backend == "nccl" else self.backend + " (host-staged rehearsal)") if self.on else "none (single process)"

    def close(self):
        if self.on:
            self.dist.barrier()
            self.dist.destroy_process_group()


