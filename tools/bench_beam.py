"""Beam-search timing (prune pre-pass + search), 32 x 249 frames, V = 4233, top-40 pruning, per posterior kind:
  flat     synthetic logits x 3 (random-init-model-like: ~40 candidates survive the pruning of every frame)
  peaky    one dominant character per frame
  trained  blank-dominated frames with short character spikes (what a trained CTC model emits: most frames keep 1 - 3
           candidates)
Lines: beam 10 / 16 / 100 / 300 without a scorer (beams <= 16 also with PPASR_BEAM_FAST=0 = no restricted list), and
beam 10 / 300 with a character 3-gram scorer (alpha 2.2, beta 4.3: the reference's shipped decode configuration,
configs/conformer.yml:78-92).  `python tools/bench_beam.py [flat] [peaky] [trained] [--quick]`"""
import os, sys, tempfile, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lm_util import write_synthetic_arpa
from ppasr_amd.decoders.beam_search_decoder import Scorer, beam_search_ids
B, T, V = 32, 249, 4233
kinds = [k for k in ("flat", "peaky", "trained") if k in sys.argv] or ["flat", "trained"]
vocab = ["<blank>", "<unk>"] + [chr(0x4E00 + i) for i in range(V - 3)] + ["<eos>"]
rng = np.random.Generator(np.random.PCG64(0))
known = [c for c in vocab[2:-1] if rng.random() < 0.8]
arpa = write_synthetic_arpa(os.path.join(tempfile.mkdtemp(), "lm.arpa"), known, order=3, n_sent=3000, sent_len=24, seed=17)
scorer = Scorer(2.2, 4.3, arpa, vocab)


def table(kind):
    logits = rng.standard_normal((B, T, V)).astype(np.float32)
    if kind == "flat":
        logits *= 3
    elif kind == "peaky":
        logits *= 3
        idx = rng.integers(0, V, size=(B, T))
        np.put_along_axis(logits, idx[..., None], 14.0, axis=-1)
    else:
        for b in range(B):
            t = 0
            while t < T:
                if rng.random() < 0.55:
                    n = int(rng.integers(1, 6)); logits[b, t:t + n, 0] += 15.0
                else:
                    n = int(rng.integers(1, 4)); logits[b, t:t + n, int(rng.integers(1, V))] += 14.0
                    for alt in rng.integers(1, V, size=3):
                        logits[b, t:t + n, alt] += float(rng.uniform(8.0, 13.0))
                    if rng.random() < 0.5:
                        logits[b, t:t + n, 0] += 12.0
                t += n
    return torch.softmax(torch.from_numpy(logits), -1).cuda()


def run(p, beam, env, lm):
    for k in ("PPASR_BEAM_FAST", "PPASR_BEAM_WAVE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    beam_search_ids(p, beam, 0.99, 40, 0, ext_scorer=lm); torch.cuda.synchronize()
    reps = 3 if "--quick" in sys.argv else 5
    t = time.perf_counter()
    for _ in range(reps): beam_search_ids(p, beam, 0.99, 40, 0, ext_scorer=lm)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


for kind in kinds:
    p = table(kind)
    for beam in (10, 16, 100, 300):
        modes = (("default", {}), ("full rows", {"PPASR_BEAM_FAST": "0"})) if beam <= 16 or "--full" in sys.argv else (("default", {}),)
        for name, env in modes:
            dt = run(p, beam, env, None)
            print(f"[{kind}] beam {beam} ({name}): {dt*1e3:.2f} ms per batch of {B} x {T} frames = {dt/T*1e6:.2f} us/frame", flush=True)
    for beam in (10, 300):
        dt = run(p, beam, {}, scorer)
        print(f"[{kind}] beam {beam} + 3-gram scorer (alpha 2.2, beta 4.3): {dt*1e3:.2f} ms per batch of {B} x {T} frames = {dt/T*1e6:.2f} us/frame", flush=True)
for k in ("PPASR_BEAM_FAST", "PPASR_BEAM_WAVE"):
    os.environ.pop(k, None)
