"""Beam-search timing (prune pre-pass + search), 32 x 249 frames, V = 4233, top-40 pruning.  Beams <= 16 are timed on both
routes: the block-wide kernel (PPASR_BEAM_WAVE=0) and the one-wave-per-utterance kernel (default)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
rng = np.random.Generator(np.random.PCG64(0))
B, T, V = 32, 249, 4233
logits = rng.standard_normal((B, T, V)).astype(np.float32) * 3
p = torch.softmax(torch.from_numpy(logits), -1).cuda()
for beam in (10, 16, 100, 300):
    for wave in ((0, 1) if beam <= 16 else (0,)):
        os.environ["PPASR_BEAM_WAVE"] = str(wave)
        beam_search_ids(p, beam, 0.99, 40, 0); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5): beam_search_ids(p, beam, 0.99, 40, 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        print(f"beam {beam} ({'wave' if wave else 'block'} kernel): {dt*1e3:.2f} ms per batch of {B} x {T} frames = {dt/T*1e6:.2f} us/frame", flush=True)
os.environ.pop("PPASR_BEAM_WAVE", None)
