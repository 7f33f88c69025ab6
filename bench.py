#!/usr/bin/env python
"""bench.py -- audio-seconds/s of PPASR's encoder-forward + CTC-decode hot path on MI355X.

Default workload (BASELINE.json configs[1]): configs/conformer.yml, 32 synthetic 10 s utterances
(1000 fbank frames x 80) PER GPU, random-init weights with the reference's initialisers, V = 4233, ctc_greedy.  One
"step" = one pass of the hot path over one batch: features resident in HBM -> token ids + scores on device (+ one
all-gather of the packed hypotheses when N > 1).  Utterances are sharded across ranks (weak scaling, no data-path
collective other than that gather).

    python bench.py                       # N = 1, configs[1]
    python bench.py --config cfg4         # the other BASELINE configs (parity-test cases with the same evidence):
                                          #   cfg1 DeepSpeech2 non-streaming B=1 5 s greedy
                                          #   cfg3 = cfg2 at --gpus 8 (global batch 256)
                                          #   cfg4 Efficient-Conformer B=64, ctc_beam_search beam 10
                                          #   cfg5 Squeezeformer, 16 utterances of 2-30 s per GPU (global B = 16 N),
                                          #        length buckets dealt to the ranks, ragged encode, beam 10
    python bench.py --gpus 8              # starts its own 8 ranks (re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 ...             # the driver's form: used as is
    python bench.py --gpus 2 --dry-run-cpu   # plumbing check without a GPU: gloo, 2 ranks, stub kernels (value is null)

Prints ONE JSON line on rank 0 (contract in the task statement): `value` comes from the wall time of EXACTLY K steps
bracketed by barrier + synchronize (max over ranks).  Extra fields:
  median_ms_per_step - median of the K per-step durations measured with HIP events on the launch stream
  n_ranks_seen       - torch.distributed world size as RCCL saw it (asserted == --gpus), `backend`
  roofline           - dominant kernel's algorithmic FLOP/s (or bytes/s) vs the chip peak; its launch duration is measured
                       live with HIP events attached to every dispatch (ppasr_kprof_* in the C-ABI); `traffic` = HBM bytes
                       per launch from the committed rocprofv3 PMC passes, used only when they were collected for THIS
                       build of the kernels
  cpu_baseline       - the torch-CPU oracle (+ the reference-style decoder) timed on this host's cores (rank 0, N=1)
"""
import argparse
import glob
import hashlib
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E
# --gemm f16x3 (opt-in mode, csrc/h3.h): one fp32 product = three fp16 MFMA products, so the mode's own denominator is the
# dense fp16 matrix peak (2.5 PFLOP/s, MI355X_MICROARCH.md) / 3, in fp32-equivalent FLOPs
F16X3_PEAK_TFLOPS = 2500.0 / 3.0
FRAME_SHIFT_S = 0.010
D, FF, F_IN, V_DEFAULT = 256, 2048, 80, 4233
BEAM = dict(beam_size=10, cutoff_prob=0.99, cutoff_top_n=40)


# ---------------------------------------------------------------------------------------------------------------------
# Algorithmic work (MAC = 2 FLOP).  Weight-only folding (linear_pos(pos_emb)) is not counted.
# ---------------------------------------------------------------------------------------------------------------------
def front_dims(T, F=F_IN):
    t1, f1 = (T - 1) // 2, (F - 1) // 2
    return t1, f1, (t1 - 1) // 2, (f1 - 1) // 2


def conformer_flops_per_utt(T, F=F_IN, d=D, ff=FF, L=12, k=15, V=V_DEFAULT):
    """Algorithmic FLOPs per utterance of ONE launch of each kernel class of the Conformer route."""
    t1, f1, tp, f2 = front_dims(T, F)
    s1 = (4 * d * ff + 6 * d * d) * tp            # one k_ffn_qkv launch (FFN_macaron + QKV)
    s4 = (2 * k * d + 2 * d * d + 4 * d * ff) * tp  # one k_conv_ffn launch (dwconv + pw2 + FFN)
    per = {
        "k_conv1": 2 * 9 * d * t1 * f1,
        "k_gemm_stream<conv2>": 2 * 9 * d * d * tp * f2,
        "k_gemm_stream<embed>": 2 * (d * f2) * d * tp,
        "k_ffn_qkv": s1,
        "k_attention": 6 * tp * tp * d,
        "k_out_glu": (2 * d * d + 4 * d * d) * tp,
        "k_attn_out_glu": 6 * tp * tp * d + (2 * d * d + 4 * d * d) * tp,
        "k_conv_ffn": s4,
        "k_conv_ffn+ffn_qkv": s4 + s1,
        "k_ctc_head": 2 * d * V * tp,
    }
    return per, tp


def class_of(name):
    """Kernel name (as rocprofv3 prints it, parameter list dropped) -> accounting class."""
    n = name
    if n.startswith("k_gemm_stream<"):
        return "conv2" if "Conv2S" in n else "dense"   # (rocprof summaries may truncate the name)
    m = re.match(r"(k_[a-z0-9_]+)(<[^>]*>)?", n)
    if not m:
        return n
    base, targs = m.group(1), m.group(2) or ""
    # round-4 kernels on transposed accumulators (csrc/rbt.h, attention_kernels.hip): the classes of the kernels they
    # replace -- k_sq_mid_t<R> -> k_sq_mid, k_sq_tail_t<R, KS> -> k_sq_tail<KS>, k_attention_t<DK> -> k_attention<DK>
    if base == "k_sq_mid_t":
        return "k_sq_mid"
    if base == "k_sq_tail_t":
        return "k_sq_tail<%s>" % targs.strip("<>").split(",")[-1].strip()
    if base == "k_attention_t":
        return "k_attention<%s>" % targs.strip("<>").split(",")[0].strip()
    # ... and the Conformer-family kernels on the 16-row / 16-wave block forms (conformer_kernels_t.hip):
    # k_conv_ffn_t<R, KS, NEXT> -> k_conv_ffn<KS>[+next], k_ffn_qkv_t<R> -> k_ffn_qkv, k_out_glu_t<R> -> k_out_glu
    if base == "k_conv_ffn_t":
        a = [t.strip() for t in targs.strip("<>").split(",")]
        return f"k_conv_ffn<{a[1]}>+next" if a[-1] in ("true", "1") else f"k_conv_ffn<{a[1]}>"
    if base in ("k_ffn_qkv_t", "k_out_glu_t"):
        return base[:-2]
    # the opt-in fp16 x3 instantiations (ppasr_set_gemm_mode, csrc/h3.h) keep classes of their own -- "<class of the
    # fp32 kernel>/f16x3" -- so that a trace holding both modes (bench.py runs the mode's steps after the timed region)
    # never averages the two: k_conv_ffn_h3<KS, NEXT>, k_ffn_qkv_h3, k_attn_out_glu_h3, k_sq_mid_h3, k_sq_tail_h3<KS>,
    # k_conv_stage_h3<MT>, k_embed_h3<SB>, k_ctc_head_h3<LOGITS>
    if base.endswith("_h3"):
        a = [t.strip() for t in targs.strip("<>").split(",")] if targs else []
        if base == "k_conv_ffn_h3":
            c = f"k_conv_ffn<{a[0]}>+next" if a[-1] in ("true", "1") else f"k_conv_ffn<{a[0]}>"
        elif base == "k_sq_tail_h3":
            c = f"k_sq_tail<{a[0]}>"
        elif base == "k_conv_stage_h3":
            c = "conv2"
        elif base == "k_embed_h3":
            c = "dense"
        else:
            c = base[:-3]
        return c + "/f16x3"
    if base == "k_conv_ffn":
        a = [t.strip() for t in targs.strip("<>").split(",")]
        return f"k_conv_ffn<{a[0]}>+next" if a[-1] in ("true", "1") else f"k_conv_ffn<{a[0]}>"
    if base in ("k_attention", "k_conv_ffn_stride", "k_sq_tail", "k_conv_pre", "k_ctc_beam"):
        a = targs.strip("<>").split(",")[0].strip()
        return f"{base}<{a}>"
    return base


def former_class_flops(family, utt_frames, padded_T, L=12, V=V_DEFAULT, d=D, ff=FF):
    """-> {class: algorithmic FLOPs per step} for a batch of utterances with `utt_frames` valid input frames each.
    Rows of an utterance = its valid encoder frames ceil(len / 4) (capped by the batch's output frames); the front end
    is counted over the same valid frames.  Mirrors the launch sequence of csrc/capi.hip / capi_squeezeformer.hip for
    the fused routes; classes of the split route for under-filled launches get their share of the same sums."""
    _, f1, Tp_pad, f2 = front_dims(padded_T)
    fl = {}

    def add(c, v):
        fl[c] = fl.get(c, 0) + int(v)
    ffn, qkv, outp, pw1, pw2 = 4 * d * ff, 6 * d * d, 2 * d * d, 4 * d * d, 2 * d * d
    for ln in utt_frames:
        tp = min((int(ln) + 3) // 4, Tp_pad)
        t1 = min((int(ln) + 1) // 2, (padded_T - 1) // 2)
        add("k_conv1", 2 * 9 * d * t1 * f1)
        add("conv2", 2 * 9 * d * d * tp * f2)
        add("dense", 2 * (d * f2) * d * tp)
        if family == "conformer":
            k = 15
            add("k_ffn_qkv", (ffn + qkv) * tp)
            add("k_attn_out_glu", L * (6 * tp * tp * d + (outp + pw1) * tp))
            add(f"k_conv_ffn<{k}>+next", (L - 1) * (2 * k * d + pw2 + ffn + ffn + qkv) * tp)
            add(f"k_conv_ffn<{k}>", (2 * k * d + pw2 + ffn) * tp)
            add("k_ctc_head", 2 * d * V * tp)
        elif family == "efficient_conformer":
            # layers 0-3 grouped attention (3 frames per token, head width 192), layer 3 = stride-2 depthwise conv,
            # layers 4-11 at half rate with 7-tap kernels
            tt, th = (tp + 2) // 3, (tp + 1) // 2
            add("k_ffn_qkv", (ffn + qkv) * tp + (ffn + qkv) * th)          # layer 0, and layer 4 behind the stride layer
            add("k_attention<192>", 4 * 6 * tt * tt * 3 * d)
            add("k_out_glu", 4 * (outp + pw1) * tp)
            add("k_conv_ffn<15>+next", 3 * (2 * 15 * d + pw2 + ffn + ffn + qkv) * tp)
            add("k_conv_ffn_stride<15>", (2 * 15 * d + pw2 + ffn) * th)
            add("k_attn_out_glu", 8 * (6 * th * th * d + (outp + pw1) * th))
            add("k_conv_ffn<7>+next", 7 * (2 * 7 * d + pw2 + ffn + ffn + qkv) * th)
            add("k_conv_ffn<7>", (2 * 7 * d + pw2 + ffn) * th)
            add("k_ctc_head", 2 * d * V * th)
        elif family == "squeezeformer":
            # post-LN blocks MHA -> FFN -> conv(31) -> FFN; layers 5..10 at half rate, recovery before layer 11
            th = (tp + 1) // 2
            full, half = 6, 6
            add("k_sq_qkv", qkv * tp)
            add("k_attention<64>", full * 6 * tp * tp * d + half * 6 * th * th * d)
            add("k_sq_mid", (outp + ffn + pw1) * (full * tp + half * th))
            # tail of layer i carries the QKV of layer i+1 except where reduce / recover produce it and after the last
            add("k_sq_tail<31>", (2 * 31 * d + pw2 + ffn) * (full * tp + half * th) + qkv * (4 * tp + 5 * th))
            add("k_sq_reduce", (2 * d * d + qkv) * th)
            add("k_sq_recover", (2 * d * d + qkv) * tp)
            add("k_ctc_head", 2 * d * V * tp)
    return fl


# classes whose work the split route for under-filled launches takes over from a fused class (same FLOPs, other kernels)
SPLIT_ALIASES = {
    "k_sq_mid": ("k_sq_oproj", "k_ffn_part", "k_ffn_join", "k_sq_pw1glu"),
    "k_sq_tail<31>": ("k_conv_pre<31>", "k_ffn_part", "k_ffn_join", "k_sq_qkv"),
}


def csrc_digest():
    """sha256 over the kernel sources: stamps PMC evidence (profiles/hbm_traffic*.json) to the build it was taken on.
    (.git does not travel to the GPU box, so the git head cannot be the stamp.)"""
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "ppasr_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "ppasr_amd", "csrc", "*.h"))):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm", default="f32", choices=["f32", "f16x3"],
                    help="f16x3: the WHOLE line in the opt-in fp16 x 3 GEMM mode (ppasr_set_gemm_mode; cfg2 / cfg4 / cfg5), dtype "
                         "'f16x3-split', its own roofline denominator; never the headline (default f32: exact fp32 MFMA)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="cfg4 / cfg5: run the beam search of a step on the encoder's stream instead of overlapping it "
                         "with the next step's encoder")
    ap.add_argument("--ragged-mode", default="merged", choices=["merged", "buckets"], help="cfg5: encoder batches per rank")
    ap.add_argument("--share-gpu", action="store_true",
                    help="N > 1 on a box with fewer GPUs: the ranks share the visible devices, gloo backend with host-staged "
                         "collectives; the real workload and the whole multi-rank flow, NOT a measurement")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="no GPU: gloo backend and a stub step; exercises launch, sharding, gather and timing plumbing")
    a = ap.parse_args()
    if a.gpus is None:
        a.gpus = 8 if a.config == "cfg3" else 1
    defaults = {"cfg1": (200, 10, 1, 498), "cfg2": (300, 10, 32, 1000), "cfg3": (300, 10, 32, 1000),
                "cfg4": (100, 5, 64, 1000), "cfg5": (100, 5, 16, None)}[a.config]
    a.steps = defaults[0] if a.steps is None else a.steps
    a.warmup = defaults[1] if a.warmup is None else a.warmup
    a.batch = defaults[2] if a.batch is None else a.batch
    a.frames = defaults[3] if a.frames is None else a.frames
    return a


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run and
    pass their output through (rank 0 prints the JSON line)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    if not any(a.startswith("--gpus") for a in sys.argv[1:]):
        cmd += ["--gpus", str(args.gpus)]
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------------
# Workloads
# ---------------------------------------------------------------------------------------------------------------------
class Workload:
    """One BASELINE config on one rank: `step()` enqueues one pass and returns this rank's (tokens, n_tokens, score)
    device tensors (already gathered over the ranks when N > 1)."""
    family = None
    pipelined = False

    def finish(self):
        pass


class FormerGreedy(Workload):          # cfg2 / cfg3
    def __init__(self, args, device, rank, world, dist):
        import torch
        from ppasr_amd.model_utils.conformer.model import ConformerModel
        from ppasr_amd.parallel import gather_hypotheses
        from ppasr_amd.utils.synth import conformer_state_dict, synth_features
        self.family, self.L, self.V = "conformer", 12, V_DEFAULT
        self.B, self.T = args.batch, args.frames
        self.feats_np, self.lens_np = synth_features(self.B, self.T, seed=20240 + 200 + rank)
        conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=self.L, cnn_module_kernel=15)
        self.sd = conformer_state_dict(vocab_size=self.V, num_blocks=self.L, seed=1234)
        self.model = ConformerModel(80, self.V, streaming=True, encoder_conf=conf, state_dict=self.sd, device=device)
        self.feats = torch.from_numpy(self.feats_np).to(device)
        self.lens = torch.from_numpy(self.lens_np).to(device)
        self.dist, self.world, self._gather = dist, world, gather_hypotheses
        self.audio_s = self.B * self.T * FRAME_SHIFT_S
        self.utt_frames = [self.T] * self.B
        self.desc = ("configs[1]: Conformer streaming (configs/conformer.yml), fbank-80, "
                     f"{self.B} x {self.T * FRAME_SHIFT_S:.0f} s utterances per GPU, V=4233, ctc_greedy, "
                     "features resident in HBM -> token ids + scores on device")
        self.metric = "audio-seconds/s (RTF^-1) Conformer-streaming fbank, batch32 per GPU, ctc_greedy"
        self.decoder = "ctc_greedy"

    def step(self):
        out = self.model.encode_greedy(self.feats, self.lens)
        if self.world > 1:
            return self._gather(*out, self.dist)
        return out

    def expect_rows(self):
        return self.world * self.B

    def cpu_baseline(self):
        import torch
        from oracle.conformer_oracle import ConformerOracle
        from oracle.ctc_decoders_oracle import greedy_tokens
        oracle = ConformerOracle(self.sd, num_blocks=self.L)
        B, T = self.B, self.T

        def one_pass(bs, o):
            xb, lb = self.feats_np[o:o + bs], self.lens_np[o:o + bs]
            t1 = time.perf_counter()
            probs = oracle.get_encoder_out(xb, lb).numpy()
            for p in probs:
                greedy_tokens(p)
            return time.perf_counter() - t1, len(xb)

        thr = pick_threads(lambda: one_pass(2, 0), lambda: one_pass(B, 0)[0])

        def cpu_rate(bs, budget_s):
            done, t_cpu = 0, 0.0
            while t_cpu < budget_s:
                t, n = one_pass(bs, done % B)
                t_cpu += t
                done += n
            return done * T * FRAME_SHIFT_S / t_cpu, done, t_cpu

        # the reference evaluates in batches of 32 (trainer.py:592-645) and predicts single utterances; time the
        # oracle at the batch shape of the GPU workload and at a small batch, report the faster
        # ... each as the BEST of three bounded repetitions (the pool's hosts are shared: single timings of this baseline
        # swung by +-25 % between boxes in round 5)
        def best_of(bs, budget_s, reps=3):
            runs = [cpu_rate(bs, budget_s / reps) for _ in range(reps)]
            best = max(runs, key=lambda r: r[0])
            return best[0], sum(r[1] for r in runs), sum(r[2] for r in runs), [round(r[0], 1) for r in runs]

        r32, n32, t32, all32 = best_of(B, 9.0)
        r2, n2, t2, all2 = best_of(2, 4.5)
        best, bs, n, tt = (r32, B, n32, t32) if r32 >= r2 else (r2, 2, n2, t2)
        return {"value": round(best, 2), "unit": "audio-s/s", "cores": thr, "kind": "port",
                "sample": f"best of 3 repetitions; {n} utterances of the same workload in batches of {bs} ({tt:.1f} s of CPU work; "
                          f"batches of {B}: {all32}, batches of 2: {all2} audio-s/s), torch-CPU fp32 restatement of the Paddle "
                          f"reference (pinned to the reference's own source, tests/test_ref_pin_cpu.py) + numpy greedy, "
                          f"torch.get_num_threads() = {torch.get_num_threads()} = {thr} of {os.cpu_count()} host threads (fastest of "
                          f"10/32/64) on {cpu_model_name()}"}


def cpu_model_name():
    """`lscpu`'s model name of the host the baseline ran on (/proc/cpuinfo; 'unknown CPU' when it is not there)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def pick_threads(warm, timed):
    """The reference's own CPU configuration is num_threads=10 (inference_predictor.py:20,68); these GEMMs do not scale
    to a whole 2-socket host and oversubscription is ruinous, so a few candidates are tried once and the fastest used."""
    import torch
    avail = os.cpu_count() or 1
    best_thr, best_t = 1, float("inf")
    for thr in sorted({min(avail, c) for c in (10, 32, 64)}):
        torch.set_num_threads(thr)
        warm()
        t = timed()
        if t < best_t:
            best_thr, best_t = thr, t
    torch.set_num_threads(best_thr)
    return best_thr


class _BeamPipe:
    """Encoder on one HIP stream, beam search (+ gather) on a second one, linked by events only: the latency-bound beam
    search of step i (one workgroup per utterance) overlaps the encoder of step i+1.  Every step's work is complete
    when `sync()` returns, which the timed region ends with."""

    def __init__(self, device, enable):
        import torch
        self.t = torch
        self.enable = enable
        self.enc = torch.cuda.Stream(device=device) if enable else None
        self.dec = torch.cuda.Stream(device=device) if enable else None
        self.dev = device

    def run(self, encode, decode):
        t = self.t
        if not self.enable:
            return decode(encode())
        main = t.cuda.current_stream(self.dev)
        self.enc.wait_stream(main)
        with t.cuda.stream(self.enc):
            probs = encode()
            ev = t.cuda.Event()
            ev.record(self.enc)
        with t.cuda.stream(self.dec):
            self.dec.wait_event(ev)
            probs.record_stream(self.dec)
            return decode(probs)

    def sync(self):
        if self.enable:
            self.dec.synchronize()
            self.enc.synchronize()


class EfficientBeam(Workload):         # cfg4
    def __init__(self, args, device, rank, world, dist):
        import torch
        from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
        from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
        from ppasr_amd.parallel import gather_hypotheses
        from ppasr_amd.utils.synth import efficient_conformer_state_dict, synth_features
        self.family, self.L, self.V = "efficient_conformer", 12, V_DEFAULT
        self.B, self.T = args.batch, args.frames
        self.feats_np, self.lens_np = synth_features(self.B, self.T, seed=20240 + 400 + rank)
        self.sd = efficient_conformer_state_dict(vocab_size=self.V, seed=1234)
        conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12, cnn_module_kernel=15, cnn_module_norm="layer_norm",
                    efficient_conf=dict(stride_layer_idx=[3], stride=[2], group_layer_idx=[0, 1, 2, 3], group_size=3))
        self.model = EfficientConformerModel(80, self.V, streaming=True, encoder_conf=conf, state_dict=self.sd, device=device)
        self.feats = torch.from_numpy(self.feats_np).to(device)
        self.lens = torch.from_numpy(self.lens_np).to(device)
        self.dist, self.world, self._gather, self._beam = dist, world, gather_hypotheses, beam_search_ids
        self.pipe = _BeamPipe(device, not args.no_pipeline)
        self.pipelined = self.pipe.enable
        self.audio_s = self.B * self.T * FRAME_SHIFT_S
        self.utt_frames = [self.T] * self.B
        self.desc = ("configs[3]: Efficient-Conformer streaming (configs/efficient_conformer.yml), fbank-80, "
                     f"{self.B} x {self.T * FRAME_SHIFT_S:.0f} s utterances per GPU, V=4233, ctc_beam_search beam 10 / 0.99 / "
                     "top-40 (HIP prefix beam search, no LM), features resident in HBM -> token ids + scores on device")
        self.metric = "audio-seconds/s (RTF^-1) Efficient-Conformer-streaming fbank, batch64 per GPU, ctc_beam_search beam 10"
        self.decoder = "ctc_beam_search"

    def _decode(self, probs):
        tokens, n, score, _ = self._beam(probs, BEAM["beam_size"], BEAM["cutoff_prob"], BEAM["cutoff_top_n"], 0)
        out = (tokens[:, 0].contiguous(), n[:, 0].contiguous(), score[:, 0].contiguous())
        if self.world > 1:
            return self._gather(*out, self.dist)
        return out

    def step(self):
        # (next to the previous step's beam search the two-launch front end is the faster one: include/ppasr_hip.h)
        self.model.set_front_fused(0 if self.pipe.enable else -1)
        return self.pipe.run(lambda: self.model.get_encoder_out(self.feats, self.lens), self._decode)

    def finish(self):
        self.pipe.sync()

    def expect_rows(self):
        return self.world * self.B

    def cpu_baseline(self):
        import torch
        from oracle.efficient_conformer_oracle import EfficientConformerOracle
        oracle = EfficientConformerOracle(self.sd, num_blocks=self.L)
        return former_beam_cpu_baseline(self, oracle, self.feats_np, self.lens_np, bs=4)


def former_beam_cpu_baseline(w, oracle, feats_np, lens_np, bs):
    """torch-CPU oracle encoder + the C restatement of the upstream beam search (one utterance after the other, like
    the reference's decoder module with num_processes=1), on a bounded sample of the workload."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libctc_beam_oracle.so"))
    lib.ctc_beam_oracle_decode.restype = ctypes.c_int

    def one_pass(o):
        xb, lb = feats_np[o:o + bs], lens_np[o:o + bs]
        T = int(lb.max())
        t1 = time.perf_counter()
        probs = oracle.get_encoder_out(xb[:, :T], lb).numpy()
        for j, p in enumerate(probs):
            n_valid = min((int(lb[j]) + 3) // 4 if w.family != "efficient_conformer" else (int(lb[j]) + 7) // 8, p.shape[0])
            p = np.ascontiguousarray(p[:n_valid], np.float32)
            L = max(p.shape[0], 1)
            tokens = np.empty((1, L), np.int32)
            lens = np.empty(1, np.int32)
            scores = np.empty(1, np.float64)
            lib.ctc_beam_oracle_decode(p.ctypes.data_as(ctypes.c_void_p), p.shape[0], p.shape[1], BEAM["beam_size"],
                                       ctypes.c_double(BEAM["cutoff_prob"]), BEAM["cutoff_top_n"], 0, 1, L,
                                       tokens.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p),
                                       scores.ctypes.data_as(ctypes.c_void_p))
        return time.perf_counter() - t1, float(lb.sum()) * FRAME_SHIFT_S

    thr = pick_threads(lambda: one_pass(0), lambda: one_pass(0)[0])
    t_cpu, audio, n = 0.0, 0.0, 0
    B = len(lens_np)
    while t_cpu < 10.0 and n * bs < B:
        t, a = one_pass(n * bs)
        t_cpu += t
        audio += a
        n += 1
    return {"value": round(audio / t_cpu, 2), "unit": "audio-s/s", "cores": thr, "kind": "port",
            "sample": f"{n * bs} utterances ({audio:.0f} audio-s) of the same workload in batches of {bs} ({t_cpu:.1f} s of CPU "
                      f"work), torch-CPU fp32 restatement of the Paddle reference (pinned to the reference's own source) + the C "
                      f"restatement of paddlespeech_ctcdecoders' beam search (single thread), {thr} of {os.cpu_count()} host "
                      f"threads for the encoder (fastest of 10/32/64) on {cpu_model_name()}"}


class SqueezeformerRagged(Workload):   # cfg5
    def __init__(self, args, device, rank, world, dist):
        import torch
        from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
        from ppasr_amd.parallel import RaggedPlan, beam_ids_decoder
        from ppasr_amd.utils.synth import squeezeformer_state_dict, synth_features
        self.family, self.L, self.V = "squeezeformer", 12, V_DEFAULT
        per = args.batch
        self.sd = squeezeformer_state_dict(vocab_size=self.V, seed=1234)
        conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=12, reduce_idx=5, recover_idx=11,
                    feed_forward_expansion_factor=8, cnn_module_kernel=31)
        self.model = SqueezeformerModel(80, self.V, streaming=True, encoder_conf=conf, state_dict=self.sd, device=device)
        # global batch: `per` utterances per rank, lengths U{200..3000} frames; shard s = the seed of rank s in the
        # single-GPU test fixture (tests/ref_cases.py cfg5: seed 20240 + 500), shard 0 IS that fixture's batch
        lens, feats = [], []
        for s in range(world):
            rng = np.random.Generator(np.random.PCG64(20240 + 500 + s))
            ls = np.sort(rng.integers(200, 3001, size=per))[::-1].astype(np.int64)
            x, _ = synth_features(per, int(ls.max()), lens=ls, seed=20240 + 500 + s)
            lens += [int(v) for v in ls]
            feats += [x[j, :int(ls[j])] for j in range(per)]
        self.all_lens = lens
        self.feats_list = feats
        self.plan = RaggedPlan(self.model, feats, lens, dist=dist, mode=args.ragged_mode, device=device,
                               pipeline=not args.no_pipeline)
        self.pipelined = self.plan.pipeline
        self.dec = beam_ids_decoder(BEAM["beam_size"], BEAM["cutoff_prob"], BEAM["cutoff_top_n"], 0)
        self.world = world
        mine = [i for (idx, _x, _l, _fl) in self.plan.batches for i in idx.tolist()]
        self.utt_frames = [lens[i] for i in mine]
        self.padded_T = max(self.utt_frames) if mine else 0
        # weak scaling: the value counts the audio of ALL ranks' utterances; per-rank share for the roofline
        self.audio_s_global = float(sum(lens)) * FRAME_SHIFT_S
        self.audio_s = self.audio_s_global / world
        self.n_global = len(lens)
        self.desc = ("configs[4]: Squeezeformer streaming (configs/squeezeformer.yml), fbank-80, "
                     f"{per} utterances per GPU with lengths U{{200..3000}} frames (2-30 s), 200-frame length buckets dealt to "
                     f"the ranks by padded work (assign_buckets), each rank's buckets as "
                     f"{'ONE ragged batch (skip_padding)' if args.ragged_mode == 'merged' else 'one padded batch per bucket'}, "
                     "V=4233, ctc_beam_search beam 10 / 0.99 / top-40 over the valid frames, features resident in HBM -> "
                     "token ids + scores on device, one all-gather of the ragged hypotheses")
        self.metric = "audio-seconds/s (RTF^-1) Squeezeformer-streaming fbank, 16 variable-length (2-30 s) utterances per GPU, ctc_beam_search beam 10"
        self.decoder = "ctc_beam_search"

    def step(self):
        return self.plan.run(self.dec)

    def finish(self):
        self.plan.sync()

    def expect_rows(self):
        return self.n_global

    def cpu_baseline(self):
        from oracle.squeezeformer_oracle import SqueezeformerOracle
        oracle = SqueezeformerOracle(self.sd, num_blocks=self.L)
        # a bounded sample: the three shortest and one mid-length utterance of this rank, as one padded batch each pair
        order = sorted(range(len(self.utt_frames)), key=lambda i: self.utt_frames[i])
        pick = order[:3] + [order[len(order) // 2]]
        lens = np.array([self.utt_frames[i] for i in pick], np.int64)
        mine = [i for (idx, _x, _l, _fl) in self.plan.batches for i in idx.tolist()]
        T = int(lens.max())
        x = np.zeros((len(pick), T, F_IN), np.float32)
        for j, i in enumerate(pick):
            f = self.feats_list[mine[i]]
            x[j, :f.shape[0]] = f
        return former_beam_cpu_baseline(self, oracle, x, lens, bs=2)


class DeepSpeech2Greedy(Workload):     # cfg1
    def __init__(self, args, device, rank, world, dist):
        import torch
        from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decode_ids
        from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
        from ppasr_amd.parallel import gather_hypotheses
        from ppasr_amd.utils.synth import deepspeech2_state_dict, synth_features
        self.family, self.L, self.V, self.H = "deepspeech2", 5, V_DEFAULT, 1024
        self.B, self.T = args.batch, args.frames
        self.sd = deepspeech2_state_dict(vocab_size=self.V, streaming=False, seed=1234)
        self.model = DeepSpeech2Model(80, self.V, streaming=False, encoder_conf=dict(num_rnn_layers=5, rnn_size=1024),
                                      state_dict=self.sd, device=device)
        self.feats_np, self.lens_np = synth_features(self.B, self.T, seed=20240 + 100 + rank)
        self.feats = torch.from_numpy(self.feats_np).to(device)
        self.lens = torch.from_numpy(self.lens_np).to(device)
        self._greedy, self._gather, self.dist, self.world = greedy_decode_ids, gather_hypotheses, dist, world
        self.audio_s = self.B * self.T * FRAME_SHIFT_S
        self.desc = ("configs[0]: DeepSpeech2 non-streaming (bidirectional LSTM x 5, configs/deepspeech2.yml with streaming: "
                     f"False), fbank-80, {self.B} x {self.T * FRAME_SHIFT_S:.2f} s utterance(s) per GPU, V=4233, ctc_greedy, "
                     "features resident in HBM -> token ids + scores on device")
        self.metric = "audio-seconds/s (RTF^-1) DeepSpeech2 non-streaming fbank, batch1, ctc_greedy"
        self.decoder = "ctc_greedy"

    def step(self):
        probs = self.model.get_encoder_out(self.feats, self.lens)
        tokens, n, score, _, _ = self._greedy(probs)
        if self.world > 1:
            return self._gather(tokens, n, score, self.dist)
        return tokens, n, score

    def expect_rows(self):
        return self.world * self.B

    def class_work(self):
        """-> {class: (work per step, unit, bound)}: the recurrence re-reads W_hh of both directions every time step
        (algorithmic bytes; B = 1 makes it a matrix-vector product), the input projections are MFMA GEMMs."""
        Tp = ((self.T - 1) // 2 - 1) // 2
        H, dirs = self.H, 2
        w = {}
        w["k_lstm_step"] = (self.L * Tp * dirs * (4 * H * H * 4 + self.B * (4 * H + 3 * H) * 4), "bytes", "hbm")
        # persistent recurrence (B = 1): W_hh crosses HBM ONCE per layer launch (it lives in registers for the whole layer);
        # per time step the gate pre-activations in, h out (the layer output) and the 8-byte exchange granules out and back
        w["k_lstm_persist"] = (self.L * dirs * (4 * H * H * 4 + Tp * (4 * H * 4 + H * 4 + 2 * H * 8)), "bytes", "hbm")
        k_in = [608] + [2 * H] * (self.L - 1)
        w["dense"] = (sum(2 * k * 4 * H * Tp * self.B * dirs for k in k_in) + 2 * 2 * H * self.V * Tp * self.B, "flop", "mfma")
        return w

    def cpu_baseline(self):
        from oracle.ctc_decoders_oracle import greedy_tokens
        from oracle.deepspeech2_oracle import DeepSpeech2Oracle
        oracle = DeepSpeech2Oracle(self.sd, num_rnn_layers=self.L, streaming=False)

        def one_pass():
            t1 = time.perf_counter()
            probs = oracle.forward(self.feats_np[:1], self.lens_np[:1])[0]
            greedy_tokens(probs.numpy()[0])
            return time.perf_counter() - t1

        thr = pick_threads(one_pass, one_pass)
        t_cpu, n = 0.0, 0
        while t_cpu < 10.0:
            t_cpu += one_pass()
            n += 1
        return {"value": round(n * self.T * FRAME_SHIFT_S / t_cpu, 2), "unit": "audio-s/s", "cores": thr, "kind": "port",
                "sample": f"{n} passes over the same 4.98 s utterance ({t_cpu:.1f} s of CPU work), torch-CPU fp32 restatement of "
                          f"the Paddle reference (CRNNEncoder pinned to the reference's own source; the LSTM cell is the shim's) + "
                          f"numpy greedy, {thr} of {os.cpu_count()} host threads (fastest of 10/32/64)"}


class DryRun(Workload):
    def __init__(self, args, device, rank, world, dist):
        import torch
        from ppasr_amd.parallel import gather_hypotheses
        self.B, self.T = args.batch, args.frames or 1000
        Tp = ((self.T - 1) // 2 - 1) // 2
        g = torch.Generator().manual_seed(rank)
        self.stub = (torch.randint(1, V_DEFAULT, (self.B, Tp), dtype=torch.int32, generator=g),
                     torch.full((self.B,), Tp, dtype=torch.int32), torch.rand(self.B, dtype=torch.float64, generator=g))
        self.dist, self.world, self._gather = dist, world, gather_hypotheses
        self.audio_s = self.B * self.T * FRAME_SHIFT_S
        self.desc = "dry run: stub kernels"
        self.metric = "audio-seconds/s (dry run)"
        self.decoder = "stub"

    def step(self):
        if self.world > 1:
            return self._gather(*self.stub, self.dist)
        return self.stub

    def expect_rows(self):
        return self.world * self.B


# ---------------------------------------------------------------------------------------------------------------------
ROOFLINE_REPS = 5


def roofline_leg(w, ms_per_step, reps=ROOFLINE_REPS):
    """Per-kernel durations of `reps` instrumented steps (dispatch-attached HIP events, ppasr_kprof_*), anchored on the
    un-instrumented step: marker packets stretch every kernel of an instrumented pass by the same few percent, so the
    events give each kernel's SHARE of the summed kernel time and the timed region gives the step.  With the beam search
    overlapped (pipelined) the kernels of a step sum to MORE than the step; the figures below are then each kernel's own
    average duration from its events, scaled by the same stretch factor measured on the serial Conformer route."""
    import torch
    from ppasr_amd._lib import kernel_profile
    acc = {}
    for _ in range(reps):
        with kernel_profile() as kp:
            w.step()
            w.finish()
            torch.cuda.synchronize()
        for name, (ms, n) in kp.kernels.items():
            a = acc.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += n
    total_ms = sum(a[0] for a in acc.values()) / reps
    serial = not w.pipelined
    scale = (ms_per_step / total_ms) if serial else 1.0 / 1.05
    kernels, classes = {}, {}
    for name, (ms, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        launches = n / reps
        avg_ms = (ms / n) * scale
        kernels[name] = {"launches_per_step": round(launches, 2), "avg_ms": round(avg_ms, 4),
                         "avg_ms_between_markers": round(ms / n, 4), "share_of_kernel_time": round(ms / reps / total_ms, 4)}
        c = classes.setdefault(class_of(name), [0.0, 0.0])
        c[0] += ms / reps * scale
        c[1] += launches
    return kernels, classes, total_ms


def build_roofline(w, args, ms_per_step):
    kernels, classes, total_ms = roofline_leg(w, ms_per_step)
    if w.family == "deepspeech2":
        work = w.class_work()
    else:
        fl = former_class_flops(w.family, w.utt_frames, getattr(w, "padded_T", None) or w.T, L=w.L, V=w.V)
        # the split route for under-filled launches: a fused class that never ran hands its FLOPs to the kernels that ran
        for fused, parts in SPLIT_ALIASES.items():
            ran = [p for p in parts if p in classes]
            if fused in fl and ran and fused not in classes:
                fl["+".join([fused, "split"])] = fl.pop(fused)
        # the 4x front end as one launch (csrc/front_fused.hip): both convolutions' algorithmic FLOPs on the one kernel
        # (each conv1 element counted once, as in the two-launch route; the kernel recomputes it ~2.25 times)
        if "k_conv12" in classes:
            fl["k_conv12"] = fl.pop("k_conv1", 0) + fl.pop("conv2", 0)
        work = {c: (v, "flop", "mfma") for c, v in fl.items()}
        for c in classes:
            if c.startswith("k_ctc_beam") or c.startswith("k_ctc_prune"):
                # HBM-bound by definition, latency-bound in practice: algorithmic bytes = one pass over the probability
                # table (prune) / the pruned per-frame records (search)
                n_fr = sum(min((ln + 3) // 4, 10 ** 9) for ln in w.utt_frames)
                if w.family == "efficient_conformer":
                    n_fr = sum((ln + 7) // 8 for ln in w.utt_frames)
                per_frame = w.V * 4 if c.startswith("k_ctc_prune") else (2 + 2 * 40) * 4
                work[c] = (n_fr * per_frame, "bytes", "hbm")
            if c.startswith("k_softmax_row_wg"):  # logits in, probabilities out: one read and one write of every valid row
                n_fr = sum((ln + 7) // 8 if w.family == "efficient_conformer" else (ln + 3) // 4 for ln in w.utt_frames)
                work[c] = (n_fr * w.V * 4 * 2, "bytes", "hbm")
    # --gemm f16x3: the mode's kernels keep classes of their own ("<fp32 class>/f16x3") with the fp32 class's algorithmic
    # FLOPs (fp32-equivalent) over the mode's own peak
    for c in list(classes):
        if c.endswith("/f16x3") and c[:-6] in work and c not in work:
            v, unit, _b = work.pop(c[:-6]) if c[:-6] not in classes else work[c[:-6]]
            work[c] = (v, unit, "mfma16x3")
    out_classes = {}
    for c, (ms, launches) in sorted(classes.items(), key=lambda kv: -kv[1][0]):
        e = {"ms_per_step": round(ms, 4), "launches_per_step": round(launches, 2)}
        if c in work and ms > 0:
            v, unit, bound = work[c]
            if unit == "flop":
                peak = F16X3_PEAK_TFLOPS if bound == "mfma16x3" else FP32_MFMA_PEAK_TFLOPS
                e.update(tflops=round(v / (ms * 1e-3) / 1e12, 2), gflop_per_step=round(v / 1e9, 3),
                         frac=round(v / (ms * 1e-3) / 1e12 / peak, 4), bound="mfma")
                if bound == "mfma16x3":
                    e.update(peak_tflops=round(peak, 1), arithmetic="fp16 x 3")
            else:
                e.update(gbs=round(v / (ms * 1e-3) / 1e9, 1), mbytes_per_step=round(v / 1e6, 3),
                         frac=round(v / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), bound=bound)
        out_classes[c] = e
    # dominant class = the largest one on the stream that bounds the step: with the beam search of step i overlapped
    # behind the encoder of step i + 1 (pipelined cfg4 / cfg5) that is the encoder's largest kernel, not the latency-bound
    # one-workgroup-per-utterance search running beside it (reported in `classes` and `hidden_decode_ms_per_step`)
    order = list(out_classes)
    hidden = [c for c in order if w.pipelined and (c.startswith("k_ctc_beam") or c.startswith("k_ctc_prune"))]
    dom = next(c for c in order if c not in hidden)
    de = out_classes[dom]
    members = [k for k in kernels if class_of(k) == dom]
    avg_launch_ms = round(de["ms_per_step"] / max(de["launches_per_step"], 1e-9), 4)
    traffic, traffic_note = None, "no PMC evidence for this build / config: run tools/collect_evidence.sh"
    rocprof_avg_ms = None  # the same kernel's average dispatch duration in the rocprofv3 kernel trace of the evidence pass
    tname = "hbm_traffic.json" if args.config in ("cfg2", "cfg3") else f"hbm_traffic_{args.config}.json"
    try:
        with open(os.path.join(ROOT, "profiles", tname)) as f:
            ev = json.load(f)
        if ev.get("csrc_sha256") == csrc_digest():
            if isinstance(ev.get(dom), dict):
                traffic = ev[dom].get("hbm_bytes_per_launch")
                traffic_note = ev.get("_source")
                ti = ev[dom].get("hbm_bytes_per_launch_instances") or {}
                tw = sum(kernels[k]["launches_per_step"] for k in members if k in ti)
                if tw > 0:  # (instances weighted by this line's launch mix, as the profiler's durations below)
                    traffic = int(sum(kernels[k]["launches_per_step"] * ti[k] for k in members if k in ti) / tw)
                if ev[dom].get("rocprof_avg_us") is not None:
                    rocprof_avg_ms = round(ev[dom]["rocprof_avg_us"] * 1e-3, 4)
                # the class's instances (block forms) weighted by THIS line's launch mix: the traced command also runs the
                # serial and the fp16 x3 legs, which launch another mix of the same kernels
                ri = ev[dom].get("rocprof_instances") or {}
                wsum = sum(kernels[k]["launches_per_step"] for k in members if k in ri)
                if wsum > 0:
                    rocprof_avg_ms = round(sum(kernels[k]["launches_per_step"] * ri[k][1] for k in members if k in ri) / wsum * 1e-3, 4)
        else:
            traffic_note = (f"profiles/{tname} was collected on kernels {ev.get('csrc_sha256')}, this build is "
                            f"{csrc_digest()}: re-run tools/collect_evidence.sh")
    except OSError:
        pass
    whole = sum(v for c, (v, unit, _b) in work.items() if unit == "flop" and (c in classes or c.endswith("+split")))
    r = {"bound": de.get("bound", "mfma"), "kernel": dom, "kernel_instances": members}
    if de.get("bound") == "hbm":
        r.update(achieved=de.get("gbs"), peak=HBM_PEAK_GBS, unit="GB/s", frac=de.get("frac"))
    else:
        r.update(achieved=de.get("tflops"), peak=de.get("peak_tflops", FP32_MFMA_PEAK_TFLOPS), unit="TFLOP/s", frac=de.get("frac"))
        if de.get("arithmetic"):
            r.update(peak_note="fp32-equivalent FLOPs over the dense fp16 matrix peak / 3 (three fp16 MFMA products per fp32 "
                               "product, csrc/h3.h); what bounds these kernels in practice is the L2 weight stream and the "
                               "chip's sustained 16-bit MFMA clock (NOTES 9.8)")
    r.update(traffic=traffic, traffic_unit="HBM bytes per launch (rocprofv3 PMC)", traffic_source=traffic_note,
             avg_launch_ms=avg_launch_ms, avg_launch_ms_rocprof=rocprof_avg_ms,
             avg_launch_ms_rocprof_note=("average dispatch duration of the same kernel(s) in the rocprofv3 --kernel-trace pass of "
                                         "tools/collect_evidence.sh on this build (profiles/*_kernel_trace_bench.txt), instances "
                                         "weighted by this line's launches per step; the profiler "
                                         "adds ~1 us per dispatch, which shows on 5 us kernels (cfg1) and not on 300 us ones"),
             hidden_decode_ms_per_step=(round(sum(out_classes[c]["ms_per_step"] for c in hidden), 4) if hidden else None),
             avg_launch_ms_method=("HIP-event share of the step x un-instrumented ms_per_step / launches" if not w.pipelined else
                                   "dispatch-attached HIP events / 1.05 (the marker stretch measured on the serial route).  The "
                                   "event pairs serialise the two streams, so this is the kernel's UNDISTURBED duration and `frac` "
                                   "the kernel's own efficiency; in the pipelined timed region the same kernel runs beside the "
                                   "previous step's beam search and takes longer (avg_launch_ms_rocprof, from the kernel trace of "
                                   "this command: +30 .. 45 % on cfg5's layer kernels) -- frac_in_pipeline prices it at that duration"),
             frac_in_pipeline=(round(de["frac"] * avg_launch_ms / rocprof_avg_ms, 4)
                               if (w.pipelined and rocprof_avg_ms and de.get("frac")) else None),
             whole_path_tflops_per_gpu=round(whole / (ms_per_step * 1e-3) / 1e12, 2),
             kernel_time_ms_per_step=round(total_ms, 3), classes=out_classes, kernels=kernels)
    if w.family == "deepspeech2" and dom.startswith("k_lstm_step"):
        # What the "hbm" label means here: the recurrent weights (2 x 16.8 MB per layer) are re-read on every time step and
        # stay resident in the 256 MB Infinity Cache, so `achieved` is a last-level-cache stream rate (it can exceed the
        # ~6.3 TB/s an HBM copy reaches) and the PMC traffic counts those re-reads.  Algorithmically the weights cross
        # HBM once per step of the bench.
        once = w.L * 2 * 4 * w.H * w.H * 4
        r.update(bound_note=("LLC-resident weight stream: W_hh of both directions (33.5 MB per layer) is re-read every time step "
                             "from the 256 MB Infinity Cache, not from HBM; peak stays the contract's HBM 8 TB/s for reference"),
                 weights_once_mbytes=round(once / 1e6, 1),
                 traffic_per_step_over_weights_once=(round(traffic * de["launches_per_step"] / once, 1) if traffic else None))
    if w.family == "deepspeech2" and dom.startswith("k_lstm_persist"):
        once = w.L * 2 * 4 * w.H * w.H * 4
        Tp = ((w.T - 1) // 2 - 1) // 2
        r.update(bound_note=("latency chain, not a bandwidth stream: W_hh stays in registers for the whole layer (it crosses HBM once "
                             "per launch), and the launch is T' dependent time steps, each ending in one exchange of the new h "
                             "(8 KB of tagged 8-byte granules) between the 256 workgroups -- the step time is that exchange's round "
                             "trip (MI355X_MICROARCH.md 'allgather': 2.4 - 3 us), so `frac` against the HBM peak only says how "
                             "far from bandwidth-bound the recurrence of ONE utterance is"),
                 us_per_time_step=round(avg_launch_ms * 1e3 / max(Tp, 1), 2),
                 weights_once_mbytes=round(once / 1e6, 1),
                 traffic_per_step_over_weights_once=(round(traffic * de["launches_per_step"] / once, 2) if traffic else None))
    return r


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    dry = args.dry_run_cpu
    backend = "gloo" if (dry or args.share_gpu) else "nccl"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # the collective library must have seen every rank: a silent single-rank run would report N x the work
        assert dist.get_world_size() == args.gpus and dist.get_backend() == backend, (dist.get_world_size(), dist.get_backend())
    n_ranks_seen = dist.get_world_size() if dist else 1
    device = torch.device("cpu") if dry else torch.device(
        "cuda", local_rank % torch.cuda.device_count() if args.share_gpu else local_rank)
    red_device = device if backend == "nccl" else torch.device("cpu")  # where the timing reductions live
    if not dry:
        torch.cuda.set_device(device)

    cls = DryRun if dry else {"cfg1": DeepSpeech2Greedy, "cfg2": FormerGreedy, "cfg3": FormerGreedy, "cfg4": EfficientBeam,
                              "cfg5": SqueezeformerRagged}[args.config]
    w = cls(args, device, rank, world, dist)
    if args.gemm == "f16x3" and not dry:
        if args.config not in ("cfg2", "cfg3", "cfg4", "cfg5"):
            raise SystemExit("--gemm f16x3: built for the *former configs (cfg2 .. cfg5)")
        w.model.set_gemm_mode("f16x3")

    def sync():
        w.finish()
        if not dry:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = w.step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    # per-step HIP events: on the stream the step's LAST kernel is launched on (the decode stream of a pipelined
    # workload, else torch's current stream = the stream the C-ABI launches on)
    ev_stream = None
    if not dry:
        ev_stream = getattr(getattr(w, "plan", None), "dec_stream", None) or getattr(getattr(w, "pipe", None), "dec", None) \
            or torch.cuda.current_stream(device)
    events = None if dry else [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    if events:
        events[0].record(ev_stream)
    for i in range(args.steps):
        out = w.step()
        if events:
            events[i + 1].record(ev_stream)
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the gathered batch really holds every rank's utterances
    assert out[0].shape[0] == w.expect_rows(), (out[0].shape, w.expect_rows())
    ms_per_step = elapsed / args.steps * 1e3
    audio_s_per_step = getattr(w, "audio_s_global", None) or world * w.audio_s
    value = audio_s_per_step / (elapsed / args.steps)
    median_ms = None
    if events:
        per = np.array([events[i].elapsed_time(events[i + 1]) for i in range(args.steps)])
        median_ms = float(np.median(per))
        if world > 1:
            t = torch.tensor([median_ms], dtype=torch.float64, device=red_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            median_ms = float(t.item())

    # Pipelined workloads (cfg4 / cfg5): the same steps once more with the overlap switched off -- encoder, then its beam
    # search, then the next step -- so that the line carries BOTH figures: `value` = the pipelined rate (a caller with a
    # stream of batches), config.serial = what a caller with ONE batch at a time gets.  Every rank runs it (N > 1: a
    # step ends in the all-gather).
    serial = None
    if w.pipelined and not dry:
        holder, attr = (w.plan, "pipeline") if hasattr(w, "plan") else (w.pipe, "enable")
        setattr(holder, attr, False)
        try:
            n_ser = max(5, min(args.steps, 30))
            for _ in range(2):
                w.step()
            sync()
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            for _ in range(n_ser):
                w.step()
            sync()
            if world > 1:
                dist.barrier()
            el = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([el], dtype=torch.float64, device=red_device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            serial = {"steps": n_ser, "ms_per_step": round(el / n_ser * 1e3, 3), "value": round(audio_s_per_step / (el / n_ser), 1),
                      "unit": "audio-s/s", "note": "same workload, encoder and beam search of a step back to back on one stream"}
        finally:
            setattr(holder, attr, True)

    # The roofline leg runs `reps` more steps of the workload; with N > 1 a step ends in the hypothesis all-gather, so
    # EVERY rank runs the leg (each profiles its own kernels; rank 0's figures are reported) -- a rank-0-only leg would
    # leave rank 0 waiting in a collective the other ranks never enter.  The dry run mirrors the extra steps.
    roofline = None
    if dry:
        for _ in range(ROOFLINE_REPS):
            w.step()
        w.finish()
    else:
        roofline = build_roofline(w, args, ms_per_step)
        if rank != 0:
            roofline = None
    if world > 1:
        dist.barrier()

    # cfg2 / cfg4 / cfg5: the same steps once more with the large GEMMs on the opt-in fp16 x3 route (ppasr_set_gemm_mode, csrc/h3.h).
    # NOT the headline: `value` above is the default fp32-MFMA mode; this figure goes to config.f16x3 together with whether
    # the token ids of the batch (greedy / beam search) are the default mode's.  Every rank runs it (a step ends in the
    # all-gather).
    f16x3 = None
    if not dry and args.config in ("cfg2", "cfg4", "cfg5") and args.gemm == "f32":
        ref_out = w.step()
        sync()
        ref_ids = [t.clone() for t in ref_out[:2]]
        sync()
        try:
            w.model.set_gemm_mode("f16x3")
            n_h = max(5, min(args.steps, 100))
            for _ in range(3):
                w.step()
            sync()
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            for _ in range(n_h):
                w.step()
            sync()
            if world > 1:
                dist.barrier()
            el = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([el], dtype=torch.float64, device=red_device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            got = w.step()
            sync()
            # per utterance: same token count and the same tokens (a beam search prunes on a cumulative-probability
            # threshold, so a 1e-6 change of the probabilities can change a candidate set and with it a hypothesis)
            rt, rn, gt, gn = ref_ids[0].cpu(), ref_ids[1].cpu(), got[0].cpu(), got[1].cpu()
            same = sum(int(rn[i] == gn[i] and bool(torch.equal(rt[i, :int(rn[i])], gt[i, :int(gn[i])])))
                       for i in range(rt.shape[0]))
            f16x3 = {"steps": n_h, "ms_per_step": round(el / n_h * 1e3, 3), "value": round(audio_s_per_step / (el / n_h), 1),
                     "unit": "audio-s/s", "utterances_with_the_default_modes_tokens": f"{same} of {rt.shape[0]}",
                     "note": "opt-in mode, not the headline: " + ("the feed-forward GEMMs of the 32-row layer kernels and conv2 of the "
                                                                   "front end" if args.config == "cfg5" else
                                                                   "the feed-forward GEMMs of the layer kernels and conv2 of the front end") +
                             " as three fp16 MFMAs per 16-wide k step on two-piece operands (22 significant bits, exact "
                             "products, fp32 accumulation; NOTES.md 9.8); everything else unchanged"}
        except Exception as e:  # (a build without the mode: report, do not fail the headline)
            f16x3 = {"error": str(e)[:200]}
        finally:
            w.model.set_gemm_mode("f32")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dry:
        cpu = w.cpu_baseline()

    if rank == 0:
        cfg_name = {"cfg1": "configs[0]", "cfg2": "configs[1]", "cfg3": "configs[2]", "cfg4": "configs[3]",
                    "cfg5": "configs[4]"}[args.config]
        line = {
            "metric": w.metric,
            "value": None if dry else round(value, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": ("f16x3-split" if args.gemm == "f16x3" else "f32"), "data": ("dry-run: stub kernels, plumbing only" if dry else
                                         "synthetic (ranks SHARE devices, host-staged gloo collectives: plumbing check, not a measurement)"
                                         if args.share_gpu else "synthetic"),
            "config": {"workload": w.desc, "baseline_config": cfg_name,
                       "global_batch": getattr(w, "n_global", None) or world * args.batch, "frames": args.frames,
                       "decoder": w.decoder, "parallelism": f"utterance-dp{world}",
                       "pipelined": bool(w.pipelined), "serial": serial, "f16x3": f16x3, "gemm": args.gemm,
                       "pipelined_note": ("the beam search of step i runs on a second HIP stream and overlaps the encoder of step "
                                          "i+1; all K steps are complete when the timed region ends") if w.pipelined else None},
            "median_ms_per_step": None if median_ms is None else round(median_ms, 3),
            "value_from_median": None if median_ms is None else round(audio_s_per_step / (median_ms * 1e-3), 1),
            "timed_region_s": round(elapsed, 3), "n_ranks_seen": n_ranks_seen, "backend": backend if world > 1 else None,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
