"""Encoder latency of ONE 10 s utterance (B = 1), the three *former families at 12 blocks (best of 5 x 20 calls)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stream_families import build  # noqa: E402  (prints its own chunk line first)
from ppasr_amd.utils.synth import synth_features
out = {"tag": os.environ.get("TAG", "")}
x, la = synth_features(1, 1000, seed=5)
for fam in ("conformer", "efficient", "squeezeformer"):
    model = build(fam)
    xd, lad = torch.from_numpy(x).cuda(), torch.as_tensor(la).cuda()
    best = 1e9
    for rep in range(6):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            model.get_encoder_out(xd, lad)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 20 * 1e3
        if rep:
            best = min(best, dt)
    out[fam + "_10s_ms"] = round(best, 3)
print(json.dumps(out), flush=True)
