#!/usr/bin/env python
"""Numerics study for a split-bf16 matrix-core route (CPU only; reads nothing but this repository).

gfx950 has no xf32 / TF32: an fp32 GEMM on the matrix cores runs v_mfma_f32_32x32x2_f32 at 1/16 of the bf16 rate.
Writing each fp32 operand as a sum of bf16 pieces (a = a0 + a1 [+ a2], every piece exactly representable) turns one fp32
product into a few bf16 MFMA products with fp32 accumulation -- every partial product a_i * b_j is EXACT in fp32
(8 x 8 mantissa bits), only the pieces that are dropped and the accumulation round.

  x3 : a = a0 + a1, b = b0 + b1,   a0 b0 + a0 b1 + a1 b0                    (3/16 of the fp32-MFMA time)
  x6 : three pieces each,          a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0   (6/16)

This script runs the Conformer oracle (oracle/conformer_oracle.py) in float64 (the "truth"), in float32 (what the HIP
kernels reproduce to 1e-6) and with every GEMM-shaped operation replaced by the split emulation, on the synthetic
BASELINE-shaped model, and prints the logit error of each against the truth plus the number of greedy frames that change.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", ".."))
from oracle import conformer_oracle as co  # noqa: E402
from ppasr_amd.utils import synth  # noqa: E402


def pieces(x, n, trunc=False, half=False):
    out, r = [], x
    if half:  # fp16 pieces, round to nearest (11 significant bits each; subnormals below 2^-14 keep 2^-24 absolute)
        for _ in range(n):
            p = r.to(torch.float16).to(torch.float32)
            out.append(p)
            r = r - p
        return out
    for _ in range(n):
        if trunc:  # piece = the top 16 bits (what `a & 0xffff0000` gives on the device)
            p = (r.view(torch.int32) & -65536).view(torch.float32)
        else:
            p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    return out


TERMS = {"x3": (2, [(0, 0), (0, 1), (1, 0)]),
         "x4": (2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
         "x6": (3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]),
         "x1": (1, [(0, 0)]),
         # fp16 pieces: 11 + 11 bits = 2^-22 per operand, a product of two pieces is still exact in fp32 (22 bits).  The
         # second operand (weights; keys / values in attention) is pre-scaled by 2^8 so that its low piece stays a normal
         # fp16 number (weights of 0.05 have low pieces of 2e-5, fp16's smallest normal is 6e-5); 2^-8 on the result is exact
         "h3": (2, [(0, 0), (0, 1), (1, 0)]),
         "h4": (2, [(0, 0), (0, 1), (1, 0), (1, 1)])}


class Split:
    def __init__(self, mode, trunc=False):
        self.n, self.terms = TERMS[mode]
        self.trunc = trunc
        self.half = mode.startswith("h")
        self.overflow = 0
        self.scales = (1.0, 256.0)

    def mm(self, a, b):
        if self.half:
            sa, sb = self.scales
            self.overflow += int((a.abs() * sa > 65504).sum()) + int((b.abs() * sb > 65504).sum())
            pa, pb = pieces(a * sa, self.n, half=True), pieces(b * sb, self.n, half=True)
            acc = None
            for i, j in reversed(self.terms):
                t = _orig_matmul(pa[i], pb[j])
                acc = t if acc is None else acc + t
            return acc * (1.0 / (sa * sb))
        pa, pb = pieces(a, self.n, self.trunc), pieces(b, self.n, self.trunc)
        acc = None
        for i, j in reversed(self.terms):  # small terms first (the kernel would interleave them per k step)
            t = _orig_matmul(pa[i], pb[j])
            acc = t if acc is None else acc + t
        return acc


_orig_matmul = torch.matmul
_orig_conv1d = F.conv1d
_orig_conv2d = F.conv2d


class FProxy:
    """torch.nn.functional with the GEMM-shaped convolutions routed through the split product."""

    def __init__(self, sp):
        self.sp = sp

    def __getattr__(self, k):
        return getattr(F, k)

    def conv1d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if groups != 1 or w.shape[2] != 1:
            return _orig_conv1d(x, w, b, stride, padding, dilation, groups)  # depthwise: VALU work, stays fp32
        y = self.sp.mm(w[:, :, 0], x)  # [O, C] x [B, C, T]
        return y if b is None else y + b[None, :, None]

    def conv2d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if w.shape[1] == 1:
            return _orig_conv2d(x, w, b, stride, padding, dilation, groups)  # conv1: one input channel, VALU work
        B, C, H, W = x.shape
        O, _, kh, kw = w.shape
        s = stride if isinstance(stride, int) else stride[0]
        cols = F.unfold(x, (kh, kw), stride=s)  # [B, C*kh*kw, L]
        y = self.sp.mm(w.reshape(O, -1), cols)
        Ho, Wo = (H - kh) // s + 1, (W - kw) // s + 1
        y = y.reshape(B, O, Ho, Wo)
        return y if b is None else y + b[None, :, None, None]


def run(sd, feats, lens, dtype, mode=None, trunc=False, hscale=(1.0, 256.0), **oracle_kw):
    m = co.ConformerOracle(sd, dtype=dtype, **oracle_kw)
    x = torch.as_tensor(feats, dtype=dtype)
    if mode is None:
        _, logits = m.get_encoder_out(x, torch.as_tensor(lens), return_logits=True)
        return logits
    sp = Split(mode, trunc)
    sp.scales = hscale
    torch.Tensor.__matmul__ = lambda a, b: sp.mm(a, b)
    co.F = FProxy(sp)
    try:
        _, logits = m.get_encoder_out(x, torch.as_tensor(lens), return_logits=True)
    finally:
        torch.Tensor.__matmul__ = lambda a, b: _orig_matmul(a, b)
        co.F = F
    if sp.half:
        print(f"  ({mode}: {sp.overflow} operand values beyond the fp16 range)")
    return logits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--modes", default="x3,x6,h3,h4")
    ap.add_argument("--trunc", action="store_true")
    ap.add_argument("--hscale", default="1,256", help="powers of two on the first / second operand of the fp16 modes")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    sd = synth.conformer_state_dict(num_blocks=a.blocks)
    feats, lens = synth.synth_features(a.batch, a.frames)
    truth = run(sd, feats, lens, torch.float64, num_blocks=a.blocks)
    scale = truth.abs().max().item()
    ids_t = truth.argmax(-1)
    # how close are the two best logits of a frame?  (a frame can only flip if the error reaches half this margin)
    top2 = truth.topk(2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1])
    print(f"model: conformer {a.blocks} blocks, B={a.batch} T={a.frames} -> {truth.shape[1]} frames, V={truth.shape[2]}; "
          f"max |logit| {scale:.3f}; top-2 margin: min {margin.min().item():.3e}, "
          f"1e-4 quantile {torch.quantile(margin.flatten(), 1e-4).item():.3e}")
    rows = [("float32 (the fp32-MFMA kernels)", run(sd, feats, lens, torch.float32, num_blocks=a.blocks).double())]
    for mode in a.modes.split(","):
        rows.append((f"split {mode}" + (" trunc" if a.trunc else ""), run(sd, feats, lens, torch.float32, mode, a.trunc, tuple(float(v) for v in a.hscale.split(",")), num_blocks=a.blocks).double()))
    print(f"{'route':36s} {'max|err|/max|logit|':>20s} {'rms err / rms logit':>20s} {'greedy frames changed':>22s}")
    for name, lg in rows:
        err = (lg - truth).abs()
        rel = err.max().item() / scale
        rms = (err.pow(2).mean().sqrt() / truth.pow(2).mean().sqrt()).item()
        flips = int((lg.argmax(-1) != ids_t).sum())
        print(f"{name:36s} {rel:20.3e} {rms:20.3e} {flips:12d} of {ids_t.numel()}")


if __name__ == "__main__":
    main()
