"""Debug / timing harness of the beam kernel's staircase fast path: --build cross-compiles a copy of the library with
-DPPASR_BEAM_TS (per-phase stamps + fast-path counters printed by workgroup 0) into tools/_ts/; without arguments (GPU box)
it runs a few configurations with the fast path on / off, reports the first frame count at which they differ, and the stamps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
LIB = os.path.join(ROOT, "tools", "_ts", "libppasr_hip_beamts.so")
sys.path.insert(0, ROOT)
if "--build" in sys.argv:
    import __graft_entry__ as entry
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    entry.build_lib(lib=LIB, extra_flags=["-DPPASR_BEAM_TS"], obj_dir=os.path.join(ROOT, "build", "obj_beam_ts"), verbose=False)
    print("built", LIB)
    sys.exit(0)
if "--ts" in sys.argv:
    os.environ["PPASR_HIP_LIB"] = LIB
import numpy as np, torch
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
from test_ctc_beam_gpu import _probs, _oracle, _oracle_decode

def run(p, beam, cp, tn, fast, nbest):
    os.environ["PPASR_BEAM_FAST"] = "1" if fast else "0"
    tk, ln, sc, _ = beam_search_ids(torch.from_numpy(p).cuda(), beam, cp, tn, 0, nbest=nbest)
    torch.cuda.synchronize()
    return tk.cpu().numpy(), ln.cpu().numpy(), sc.cpu().numpy()

lib = _oracle()
for (T, V, beam, cp, tn, kind) in [(60, 700, 13, 0.9, 7, "flat"), (120, 500, 10, 0.99, 40, "peaky"), (249, 4233, 10, 0.99, 40, "flat")]:
    rng = np.random.Generator(np.random.PCG64(T * 11 + V + beam))
    p = _probs(rng, T, V, kind)[None]
    first = None
    for t in list(range(1, min(T, 40) + 1)) + [T]:
        a, b = run(p[:, :t], beam, cp, tn, True, beam), run(p[:, :t], beam, cp, tn, False, beam)
        if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])):
            first = t
            break
    print((T, V, beam, cp, tn, kind), "first differing frame count:", first, flush=True)
    if first:
        t = first
        a, b = run(p[:, :t], beam, cp, tn, True, beam), run(p[:, :t], beam, cp, tn, False, beam)
        ref = _oracle_decode(lib, p[0, :t], beam, cp, tn, 0, beam)
        for r in range(beam):
            print("  rank", r, "fast", a[0][0, r, :a[1][0, r]].tolist(), "%.6f" % a[2][0, r], "| general", b[0][0, r, :b[1][0, r]].tolist(),
                  "%.6f" % b[2][0, r], "| oracle", ref[r] if r < len(ref) else None)
        srt = np.argsort(-p[0, t - 1])[:8]
        print("  last frame top-8:", srt.tolist(), p[0, t - 1][srt].tolist())
