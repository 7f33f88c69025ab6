"""Ad-hoc wide sweep of batch shapes for the batched encoders (the committed tests/test_random_sweep_gpu.py runs a few)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_random_sweep_gpu as t
bad = 0; n = 0
for fam, fn, seed0 in (("conformer", t.test_conformer_sweep, 900), ("squeezeformer", t.test_squeezeformer_sweep, 901),
                       ("efficient", t.test_efficient_conformer_sweep, 902)):
    for (B, T, lens) in t._cases(seed0, 25, 7, 1400):
        n += 1
        try:
            fn(B, T, lens)
        except AssertionError as e:
            bad += 1
            print("FAIL", fam, B, T, lens, repr(e)[:120])
        except Exception as e:
            bad += 1
            print("ERROR", fam, B, T, lens, repr(e)[:160])
print("fuzz_encoders done:", n, "cases,", bad, "problems")
