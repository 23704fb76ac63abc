"""bench.py's launcher leg on a box without GPUs: `--gpus N` with no WORLD_SIZE must refuse (non-zero exit, nothing
started) when the node shows fewer than N devices, and a rank whose WORLD_SIZE disagrees with --gpus must refuse too."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_refuses_more_gpus_than_the_node_has():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "2"], cwd=ROOT,
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2 and "--gpus 64" in out.stderr and not out.stdout.strip()


def test_rank_refuses_a_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2"], cwd=ROOT,
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2 and "WORLD_SIZE 2" in out.stderr and not out.stdout.strip()
