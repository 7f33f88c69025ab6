"""Beam-search timing (prune pre-pass + search), 32 x 249 frames, V = 4233, top-40 pruning.  Beams <= 16 are timed with the
staircase fast path (default) and with the general selection only (PPASR_BEAM_FAST=0), and on the one-wave-per-utterance
kernel (PPASR_BEAM_WAVE=1).  `--probs flat|peaky`: synthetic logits x 3 (random-init-model-like, default) or tables with
one dominant character per frame (trained-model-like: most frames cut to a few candidates by cutoff_prob)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
rng = np.random.Generator(np.random.PCG64(0))
B, T, V = 32, 249, 4233
logits = rng.standard_normal((B, T, V)).astype(np.float32) * 3
if "peaky" in sys.argv:
    idx = rng.integers(0, V, size=(B, T))
    np.put_along_axis(logits, idx[..., None], 14.0, axis=-1)
p = torch.softmax(torch.from_numpy(logits), -1).cuda()
for beam in (10, 16, 100, 300):
    modes = (("fast", {}), ("general", {"PPASR_BEAM_FAST": "0"}), ("wave", {"PPASR_BEAM_WAVE": "1"})) if beam <= 16 else (("general", {}),)
    for name, env in modes:
        for k in ("PPASR_BEAM_FAST", "PPASR_BEAM_WAVE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        beam_search_ids(p, beam, 0.99, 40, 0); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5): beam_search_ids(p, beam, 0.99, 40, 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        print(f"beam {beam} ({name}): {dt*1e3:.2f} ms per batch of {B} x {T} frames = {dt/T*1e6:.2f} us/frame", flush=True)
for k in ("PPASR_BEAM_FAST", "PPASR_BEAM_WAVE"):
    os.environ.pop(k, None)
