"""Per-phase wall-clock stamps of the dominant kernel (k_conv_ffn<15,false,true>): builds a second copy of the library
with -DPPASR_PHASE_TS into tools/_ts/ (git-ignored, travels with gpurun), runs the bench workload through it and prints
the phase durations of one workgroup in the middle of the grid (100 MHz clock -> 10 ns resolution).

    python tools/phase_ts.py --build        # here (cross-compiles)
    python tools/phase_ts.py                # on the GPU box"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("PPASR_TS_LIB") or os.path.join(ROOT, "tools", "_ts", "libppasr_hip_ts.so")
NAMES = {0: "start", 1: "dwconv done", 2: "LN_cm+swish", 3: "pw2 + epilogue", 4: "LN_ff", 5: "FFN (16 units)",
         6: "residual epi", 7: "LN_final + store", 8: "LN_macaron (next)", 9: "FFN_macaron (16 units)", 10: "residual epi",
         11: "x1 store + LN_mha", 12: "Q unit", 13: "K unit", 14: "V unit", 15: "end"}

if "--build" in sys.argv:
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    entry.build_lib(lib=LIB, extra_flags=["-DPPASR_PHASE_TS"], obj_dir=os.path.join(ROOT, "build", "obj_phase_ts"))
    print("built", LIB)
    sys.exit(0)

os.environ["PPASR_HIP_LIB"] = LIB
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ppasr_amd import _lib  # noqa: E402
from ppasr_amd.model_utils.conformer.model import ConformerModel  # noqa: E402
from ppasr_amd.utils.synth import DEFAULT_VOCAB_SIZE, conformer_state_dict, synth_features  # noqa: E402

V, L = DEFAULT_VOCAB_SIZE, 12
conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=conformer_state_dict(vocab_size=V, num_blocks=L, seed=1234),
                       device="cuda:0")
if "--h3" in sys.argv:  # the fp16 x3 mode's kernels (k_conv_ffn_h3 / k_attn_out_glu_h3 carry the same stamps)
    model.set_gemm_mode("f16x3")
x, lens = synth_features(32, 1000, seed=20440)
x, lens = torch.from_numpy(x).cuda(), torch.from_numpy(lens).cuda()
for _ in range(5):
    model.encode_greedy(x, lens)
torch.cuda.synchronize()
lib = _lib.load()
acc = np.zeros(128)
reps = 10
for _ in range(reps):
    model.encode_greedy(x, lens)
    torch.cuda.synchronize()
    ts = (ctypes.c_longlong * 128)()
    # --t: the stamps of k_conv_ffn_t (conformer_kernels_t.hip: the 16-wave / 16-row forms; PPASR_W16=0 selects the 8-wave
    # k_conv_ffn read by the default)
    assert (lib.ppasr_debug_read_phase_ts_t if "--t" in sys.argv else lib.ppasr_debug_read_phase_ts)(ts) == 0
    t = np.array(list(ts), np.float64)
    acc[:64] += t[:64] - t[0]
    acc[64:] += t[64:] - t[64]
acc /= reps
us = acc / 100.0  # 100 MHz
print("k_conv_ffn<15,false,true> (last launch of the step), one workgroup, microseconds:")
prev = 0.0
cyc = acc[64:]  # shader-clock cycles since the first stamp
prevc = 0.0
for i in range(16):
    d_us, d_cyc = us[i] - prev, cyc[i] - prevc
    ghz = f"{d_cyc / d_us / 1e3:5.2f} GHz" if d_us > 0.05 else ""
    print(f"  {NAMES[i]:28s} +{d_us:8.2f}   (t = {us[i]:8.2f})   {ghz}")
    prev, prevc = us[i], cyc[i]
print("last FFN executed (macaron of the next layer): per hidden chunk, [W1(c+1) gemm + swish side] -> barrier wait -> [W2(c)]")
for c in range(8):
    a, b = us[16 + 2 * c], us[17 + 2 * c]
    nxt = us[16 + 2 * (c + 1)] if c < 7 else us[9]
    print(f"  chunk {c}: barrier wait {b - a:6.2f}   W2(c)+W1(c+2) {nxt - b:6.2f}")
if "--t" in sys.argv:
    sys.exit(0)

# per-workgroup start / end of the last k_conv_ffn<..,NEXT> launch: start skew, duration spread, per-XCD means
wv = (ctypes.c_longlong * 512)()
if lib.ppasr_debug_read_wave_ts(wv) == 0:
    print("per-wave stamps around the FFN chunk barrier (last FFN executed), us relative to wave 0's arrival at chunk 0:")
    t0 = wv[0]
    for c in range(8):
        arr = [(wv[(2 * c) * 8 + w] - t0) / 100.0 for w in range(8)]
        rel = [(wv[(2 * c + 1) * 8 + w] - t0) / 100.0 for w in range(8)]
        print(f"  chunk {c}: arrive " + " ".join(f"{a:7.2f}" for a in arr) + f"   | last - first {max(arr) - min(arr):5.2f}"
              f"   release first {min(rel):7.2f} (barrier latency {min(rel) - max(arr):5.2f})")
    print("k_attn_out_glu key loop, per wave (us since wave 0 entered its first sub-block): start | S MFMAs issued | softmax done | PV issued")
    t0 = wv[16 * 8]
    for sb in range(2):
        for w_ in range(8):
            v = [(wv[(16 + 4 * sb + k) * 8 + w_] - t0) / 100.0 for k in range(4)]
            print(f"  sub-block {sb} wave {w_} (head {w_ >> 1}, half {w_ & 1}): " + " ".join(f"{x:7.2f}" for x in v)
                  + f"   S {v[1] - v[0]:5.2f}  softmax {v[2] - v[1]:5.2f}  PV {v[3] - v[2]:5.2f}")
    print("k_gemm_stream<conv2, 128-row tiles>, one workgroup, per K chunk and wave (us): gemm | LDS write of the next chunk | barrier wait")
    wv = (ctypes.c_longlong * 512)()  # (conv2 lives in front_kernels.hip: that translation unit's stamps)
    assert lib.ppasr_debug_read_wave_ts_front(wv) == 0
    t0 = wv[32 * 8]
    for kc in range(8):
        rows = []
        for w_ in range(8):
            v = [(wv[(32 + 4 * kc + k) * 8 + w_] - t0) / 100.0 for k in range(4)]
            rows.append(v)
        print(f"  chunk {kc}: start " + " ".join(f"{r[0]:6.2f}" for r in rows))
        print("           gemm  " + " ".join(f"{r[1] - r[0]:6.2f}" for r in rows))
        print("           write " + " ".join(f"{r[2] - r[1]:6.2f}" for r in rows))
        print("           wait  " + " ".join(f"{r[3] - r[2]:6.2f}" for r in rows))
wg = (ctypes.c_longlong * 2048)()
assert lib.ppasr_debug_read_wg_ts(wg) == 0
w = np.array(list(wg), np.float64).reshape(1024, 2)[:249] / 100.0
t0 = w[:, 0].min()
start, end = w[:, 0] - t0, w[:, 1] - t0
dur = end - start
print(f"249 workgroups of the last NEXT launch: start skew max {start.max():.2f} us, duration min/median/max "
      f"{dur.min():.1f}/{np.median(dur):.1f}/{dur.max():.1f} us, last end {end.max():.1f} us")
for xcd in range(8):
    sel = np.arange(249) % 8 == xcd
    print(f"  XCD {xcd}: n={sel.sum():2d} start {start[sel].mean():6.2f}  duration mean {dur[sel].mean():7.1f} max {dur[sel].max():7.1f}")
order = np.argsort(-dur)[:8]
print("  slowest workgroups:", [(int(i), round(float(dur[i]), 1)) for i in order])

# ---- k_attn_out_glu: phases of one workgroup + spans of all 256
t = np.array(list(ts), np.float64)
a = (t[32:40] - t[32]) / 100.0
ac = t[96:104] - t[96]
names = ["start", "Q staging", "key loop (S, softmax, PV)", "partial O -> LDS, barrier (waits for the slowest wave)",
         "merge + bufC + barrier", "out-proj + residual epilogue + barrier", "LN_conv + barrier", "pw1 value + gate + GLU store"]
print("k_attn_out_glu (last launch), one workgroup, microseconds:")
for i in range(1, 8):
    d = a[i] - a[i - 1]
    print(f"  {names[i]:56s} +{d:7.2f}   (t = {a[i]:7.2f})   {(ac[i] - ac[i - 1]) / d / 1e3 if d > 0.05 else 0:5.2f} GHz")
w = np.array(list(wg), np.float64).reshape(1024, 2)[256:512] / 100.0
ok = w[:, 1] > w[:, 0]
w = w[ok]
t0 = w[:, 0].min()
dur = w[:, 1] - w[:, 0]
print(f"{len(w)} workgroups: start skew max {(w[:, 0] - t0).max():.2f} us, duration min/median/max {dur.min():.1f}/{np.median(dur):.1f}/{dur.max():.1f} us,"
      f" last end {(w[:, 1] - t0).max():.1f} us")
# block id -> (utterance, query block) as in the kernel's XCD-aware map: slot = id >> 3, b = (slot / nq) * 8 + (id & 7)
ids = np.nonzero(ok)[0]
nq = 8
slot = ids >> 3
bb, qb = (slot // nq) * 8 + (ids & 7), slot % nq
start = w[:, 0] - t0
for q in range(nq):
    sel = qb == q
    print(f"  query block {q}: n={sel.sum():2d} start {start[sel].mean():5.2f} duration mean {dur[sel].mean():6.2f} max {dur[sel].max():6.2f}")
for x in range(8):
    sel = (ids & 7) == x
    print(f"  XCD {x}: n={sel.sum():2d} duration mean {dur[sel].mean():6.2f} max {dur[sel].max():6.2f}   end max {(w[sel, 1] - t0).max():6.2f}")
order = np.argsort(-dur)[:10]
print("  slowest:", [(int(bb[i]), int(qb[i]), round(float(dur[i]), 1), round(float(start[i]), 2)) for i in order])
order = np.argsort(dur)[:6]
print("  fastest:", [(int(bb[i]), int(qb[i]), round(float(dur[i]), 1), round(float(start[i]), 2)) for i in order])
