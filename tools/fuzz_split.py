"""Ad-hoc sweep of the split route (ppasr_set_ffn_split) with FORCED slice counts on random batch shapes, large ones
included (the default only takes it for <= 128 row blocks): logits within 1e-5 of the fused route for every family."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_ragged_gpu as t  # noqa: E402
from ppasr_amd.utils.synth import synth_features  # noqa: E402

rng = np.random.Generator(np.random.PCG64(4321))
bad = n = 0
for family, make in t.FAMILIES.items():
    model, mul = make(211)
    for case in range(12):
        B = int(rng.integers(1, 40))
        T = int(rng.integers(7, 1400))
        lens = [int(v) for v in rng.integers(1, T + 1, size=B)]
        lens[0] = T
        x, la = synth_features(B, T, lens=lens, seed=case)
        model.set_ffn_split(0)
        ref = model.get_encoder_out(x, la, return_logits=True)[1]
        for mode in (2, 4, 8, -1):
            model.set_ffn_split(mode)
            got = model.get_encoder_out(x, la, return_logits=True)[1]
            torch.cuda.synchronize()
            err = float((got - ref).abs().max() / ref.abs().max())
            n += 1
            if not (err < 1e-5) or not bool(torch.isfinite(got).all()):
                bad += 1
                print("FAIL", family, B, T, mode, err)
    model.set_ffn_split(-1)
print("fuzz_split done:", n, "cases,", bad, "problems")
