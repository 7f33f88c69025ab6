"""`python bench.py --gpus 2 --dry-run-cpu`: the multi-rank plumbing of bench.py without a GPU -- self-launch under
torch.distributed.run, gloo, sharded seeds, the hypothesis all-gather, max-over-ranks timing, ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    line = _run([sys.executable, "bench.py", "--gpus", "2", "--dry-run-cpu", "--steps", "3", "--warmup", "1"])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["backend"] == "gloo"
    assert line["value"] is None and line["data"].startswith("dry-run")
    assert line["config"]["global_batch"] == 64 and line["steps"] == 3 and line["warmup"] == 1


def test_under_the_drivers_launcher():
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                 "127.0.0.1", "--master-port", str(29400 + os.getpid() % 500), "bench.py", "--gpus", "2", "--dry-run-cpu",
                 "--steps", "2", "--warmup", "1"])
    assert line["n_ranks_seen"] == 2 and line["scaling"] == "weak"
