"""GPU: `bench.py --gpus 2 --share-gpu` -- the REAL workload of every BASELINE config under the driver's launcher with two
ranks on the one visible GPU (gloo backend, host-staged collectives): the whole N > 1 flow (model creation per rank, the
hypothesis all-gather at the end of every step, the barrier-bracketed timing, the roofline leg on every rank, ONE JSON
line) that the dry-run test cannot reach because its steps launch nothing.  Not a measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("cfg,rows", [("cfg2", 64), ("cfg4", 128), ("cfg5", 32), ("cfg1", 2)])
def test_two_ranks_share_the_gpu(cfg, rows):
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                 "127.0.0.1", "--master-port", str(29900 + os.getpid() % 90), "bench.py", "--config", cfg, "--gpus", "2",
                 "--share-gpu", "--steps", "3", "--warmup", "1"])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["backend"] == "gloo"
    assert line["config"]["global_batch"] == rows and line["steps"] == 3
    assert "plumbing check" in line["data"] and line["value"] > 0
    assert line["roofline"] and line["roofline"]["kernel"] and line["cpu_baseline"] is None
