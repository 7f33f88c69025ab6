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


def test_every_profiled_kernel_has_an_accounting_class():
    """bench.class_of maps every ppasr kernel name of the committed rocprofv3 traces (profiles/r03c_*_kernel_trace_bench.txt,
    names possibly truncated by the summary) to an accounting class, and the classes that carry the step's matrix work
    have algorithmic FLOPs in bench.former_class_flops -- a renamed kernel must not fall out of the roofline silently."""
    import glob
    import bench
    seen = {}
    for path in glob.glob(os.path.join(ROOT, "profiles", "r03c*_kernel_trace_bench.txt")):
        for line in open(path):
            if line.startswith("#") or "|" not in line:
                continue
            name = line.split("|")[0].strip()
            if "ppasr::" not in name:
                continue
            short = name.split("ppasr::", 1)[1].split("(")[0].strip()
            seen[short] = bench.class_of(short)
    assert len(seen) >= 20, seen
    for short, cls in seen.items():
        assert cls.startswith("k_") or cls in ("conv2", "dense"), (short, cls)
    fl = bench.former_class_flops("conformer", [1000] * 32, 1000)
    for cls in ("conv2", "dense", "k_conv1", "k_ffn_qkv", "k_attn_out_glu", "k_conv_ffn<15>+next", "k_conv_ffn<15>", "k_ctc_head"):
        assert fl.get(cls, 0) > 0, cls
    assert {seen[k] for k in seen if k.startswith("k_conv_ffn<15, false, true>")} == {"k_conv_ffn<15>+next"}
    assert seen.get("k_attn_out_glu") == "k_attn_out_glu"
    # 23.28 GFLOP per 10 s utterance
    assert abs(sum(fl.values()) / 32 / 1e9 - 23.28) < 0.05, sum(fl.values()) / 32 / 1e9


def test_skip_padding_helper_on_stub_models():
    from ppasr_amd.parallel import set_skip_padding_if_built

    class Built:
        def set_skip_padding(self, on):
            self.on = on

    from ppasr_amd import _lib

    class Refuses:
        def set_skip_padding(self, on):
            raise _lib.PPASRHipError("libppasr_hip status 3: skip_padding: built for the fused 256-wide route ...",
                                     status=_lib.PPASR_EUNSUPPORTED)

    class Broken:
        def set_skip_padding(self, on):
            raise RuntimeError("hipErrorLaunchFailure")

    class OutOfMemory:  # the same words in the message, another status: must NOT be swallowed
        def set_skip_padding(self, on):
            raise _lib.PPASRHipError("libppasr_hip status 2: skip_padding: hipMalloc failed", status=_lib.PPASR_EHIP)

    b = Built()
    assert set_skip_padding_if_built(b, True) is True and b.on is True
    assert set_skip_padding_if_built(b, False) is False and b.on is False
    assert set_skip_padding_if_built(Refuses(), True) is False
    assert set_skip_padding_if_built(object(), True) is False
    import pytest
    with pytest.raises(RuntimeError):
        set_skip_padding_if_built(Broken(), True)
    with pytest.raises(_lib.PPASRHipError):
        set_skip_padding_if_built(OutOfMemory(), True)
    # check() carries the numeric status
    try:
        _lib.check(_lib.load().ppasr_set_skip_padding(None, 1))
    except _lib.PPASRHipError as e:
        assert e.status == _lib.PPASR_EINVAL
    else:
        raise AssertionError("null handle accepted")
