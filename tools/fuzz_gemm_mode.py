"""Ad-hoc sweep of the fp16 x3 GEMM mode (ppasr_set_gemm_mode, csrc/h3.h): random batch compositions for the three *former
families with the 32-row kernels forced (so that the mode's kernels run at every size), default and ragged mode; the
mode's logits must stay within 2e-5 of the default mode's, the per-frame argmax must not change outside near-ties, the
ragged mode's valid rows must be bit-identical INSIDE the mode (tests/test_gemm_mode_gpu.py runs a few fixed shapes)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_ragged_gpu as t  # noqa: E402
from ppasr_amd.utils.synth import synth_features  # noqa: E402

rng = np.random.Generator(np.random.PCG64(4242))
bad = n = skipped = 0
worst = 0.0
for family, make in t.FAMILIES.items():
    model, mul = make(211)
    try:
        model.set_gemm_mode("f16x3")
        model.set_gemm_mode("f32")
    except Exception as e:  # noqa: BLE001  (routes the mode is not built for)
        print("skip", family, repr(e)[:100])
        skipped += 1
        continue
    model.set_row_block(32)
    model.set_ffn_split(0)
    for case in range(int(os.environ.get("FUZZ_CASES", "12"))):
        B = int(rng.integers(1, 14))
        T = int(rng.integers(16, 1400))
        lens = [int(v) for v in rng.integers(1, T + 1, size=B)]
        if rng.random() < 0.7:
            lens[int(rng.integers(0, B))] = T
        x, la = synth_features(B, T, lens=lens, seed=case)
        try:
            model.set_skip_padding(False)
            model.set_gemm_mode("f32")
            _, l0 = model.get_encoder_out(x, la, return_logits=True)
            model.set_gemm_mode("f16x3")
            _, l1 = model.get_encoder_out(x, la, return_logits=True)
            model.set_skip_padding(True)
            _, l2 = model.get_encoder_out(x, la, return_logits=True)
            torch.cuda.synchronize()
            ok = bool(torch.isfinite(l1).all())
            rel = float((l1 - l0).abs().max() / l0.abs().max())
            worst = max(worst, rel)
            ok &= rel < 2e-5
            for b, ln in enumerate(lens):
                nv = min(l0.shape[1], (ln + mul - 1) // mul)
                a0, a1 = l0[b, :nv], l1[b, :nv]
                flip = a0.argmax(-1) != a1.argmax(-1)
                if bool(flip.any()):  # only acceptable where the default mode's top-2 margin is below the logit error
                    top2 = a0.topk(2, dim=-1).values
                    ok &= bool(((top2[:, 0] - top2[:, 1])[flip] < 1e-4 * float(l0.abs().max())).all())
                ok &= torch.equal(l2[b, :nv], l1[b, :nv]) and not bool(l2[b, nv:].any())
        except Exception as e:  # noqa: BLE001
            ok = False
            print("ERROR", family, B, T, lens, repr(e)[:160])
        n += 1
        if not ok:
            bad += 1
            print("FAIL", family, B, T, lens)
    model.set_skip_padding(False)
    model.set_gemm_mode("f32")
print("fuzz_gemm_mode done:", n, "cases,", bad, "problems,", skipped, "families without the mode; largest relative logit difference", worst)
