"""Ad-hoc wide sweep of the ragged-batch mode (set_skip_padding): random batch compositions for every *former family;
valid rows must be bit-identical to the default mode, everything behind them zero (tests/test_ragged_gpu.py runs a few)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_ragged_gpu as t  # noqa: E402
from ppasr_amd.utils.synth import synth_features  # noqa: E402

rng = np.random.Generator(np.random.PCG64(777))
bad = n = 0
for family, make in t.FAMILIES.items():
    model, mul = make(211)
    for case in range(25):
        B = int(rng.integers(1, 12))
        T = int(rng.integers(16, 1600))  # (>= the 8x front end's shortest utterance)
        lens = [int(v) for v in rng.integers(1, T + 1, size=B)]
        if rng.random() < 0.7:
            lens[int(rng.integers(0, B))] = T
        x, la = synth_features(B, T, lens=lens, seed=case)
        try:
            model.set_skip_padding(False)
            p0 = model.get_encoder_out(x, la)
            model.set_skip_padding(True)
            p1 = model.get_encoder_out(x, la)
            torch.cuda.synchronize()
            ok = bool(torch.isfinite(p1).all())
            for b, ln in enumerate(lens):
                nv = min(p0.shape[1], (ln + mul - 1) // mul)
                ok &= torch.equal(p0[b, :nv], p1[b, :nv]) and not bool(p1[b, nv:].any())
        except Exception as e:  # noqa: BLE001
            ok = False
            print("ERROR", family, B, T, lens, repr(e)[:160])
        n += 1
        if not ok:
            bad += 1
            print("FAIL", family, B, T, lens)
    model.set_skip_padding(False)
print("fuzz_ragged done:", n, "cases,", bad, "problems")
