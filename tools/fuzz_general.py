"""Ad-hoc wide sweep of the general Conformer layer route (csrc/capi_generic.hip): the seeded random configurations of
tests/test_general_route_gpu.py (constructor arguments x widths x front ends x ragged batches x chunking) for many more
seeds, against oracle/conformer_oracle.py.  Prints a summary."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_general_route_gpu as t  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for seed in range(14, 14 + n):
    for fn in (t.test_random_options_batched, t.test_random_options_chunked):
        try:
            fn(seed)
        except AssertionError as e:  # noqa: PERF203
            bad += 1
            print("PROBLEM", fn.__name__, seed, str(e)[:300], flush=True)
print(f"fuzz_general done: {2 * n} cases, {bad} problems")
