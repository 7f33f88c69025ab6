"""CPU: the arithmetic behind the opt-in fp16 x3 GEMM mode (csrc/h3.h, NOTES 9.8), checked on the oracle.

tools/experiments/r05/split_bf16_numerics.py replaces every GEMM of the Conformer oracle by a sum of products of 16-bit
pieces (products of pieces are exact in fp32, accumulation in fp32 -- what the matrix cores compute) and measures the logits
against the oracle in float64.  Pinned here, on a small model: the fp16 x3 split with both operands pre-scaled is in the
class of fp32 arithmetic itself, the bf16 x3 split is an order of magnitude worse (which is why the kernels use fp16 pieces),
and the piece arithmetic is exact where h3.h relies on it."""
import importlib.util
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("split_numerics", os.path.join(ROOT, "tools", "experiments", "r05",
                                                                            "split_bf16_numerics.py"))
sn = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sn)


def test_products_of_fp16_pieces_are_exact_in_fp32():
    g = torch.Generator().manual_seed(3)
    a = torch.randn(4096, generator=g) * 7.0
    b = torch.randn(4096, generator=g) * 0.05 * 256.0  # (weights pre-scaled by 2^8 as k_repack_h3 does)
    (a0, a1), (b0, b1) = sn.pieces(a, 2, half=True), sn.pieces(b, 2, half=True)
    # two pieces carry 22 significant bits of the operand ...
    assert float(((a0 + a1) - a).abs().max() / a.abs().max()) < 2.0 ** -21
    assert float(((b0 + b1) - b).abs().max() / b.abs().max()) < 2.0 ** -21
    # ... every piece is an fp16 number, and a product of two pieces (11 x 11 bits) needs no rounding in fp32
    for x in (a0, a1, b0, b1):
        assert torch.equal(x.to(torch.float16).to(torch.float32), x)
    for x, y in ((a0, b0), (a0, b1), (a1, b0)):
        assert torch.equal((x * y).double(), x.double() * y.double())
    # element by element: a relative 2^-22, or -- where the low piece falls into fp16's subnormal range -- an absolute 2^-25
    for x, (x0, x1) in ((a, (a0, a1)), (b, (b0, b1))):
        assert bool((((x0 + x1) - x).abs() <= torch.maximum(x.abs() * 2.0 ** -22, torch.tensor(2.0 ** -25))).all())
    # the 2^8 on the weights is what keeps the TYPICAL low piece (2^-11 of its value) a normal fp16 number: without it most
    # low pieces of 0.05-sized weights are subnormal
    raw1 = sn.pieces(b / 256.0, 2, half=True)[1]
    assert float((b1.abs() >= 2.0 ** -14).float().mean()) > 0.9 > float((raw1.abs() >= 2.0 ** -14).float().mean())


def test_fp16x3_logits_are_in_the_class_of_fp32_arithmetic():
    from ppasr_amd.utils import synth
    sd = synth.conformer_state_dict(num_blocks=4, vocab_size=503, seed=11)
    feats, lens = synth.synth_features(2, 400, seed=12)
    kw = dict(num_blocks=4)
    truth = sn.run(sd, feats, lens, torch.float64, **kw)
    scale = float(truth.abs().max())

    def err(lg):
        return float((lg.double() - truth).abs().max()) / scale

    e32 = err(sn.run(sd, feats, lens, torch.float32, **kw))
    eh = err(sn.run(sd, feats, lens, torch.float32, "h3", False, (16.0, 256.0), **kw))
    eb = err(sn.run(sd, feats, lens, torch.float32, "x3", **kw))
    e6 = err(sn.run(sd, feats, lens, torch.float32, "x6", **kw))
    assert e32 < 3e-6
    assert eh < 3.0 * e32 + 1e-7, (eh, e32)      # fp16 x 3 with pre-scaled operands: fp32's own class
    assert e6 < 3.0 * e32 + 1e-7, (e6, e32)      # six bf16 products likewise (6 bytes per weight: not built)
    assert eb > 3.0 * eh, (eb, eh)               # three bf16 products are not
    ids = truth.argmax(-1)
    assert np.array_equal(sn.run(sd, feats, lens, torch.float32, "h3", False, (16.0, 256.0), **kw).argmax(-1).numpy(), ids.numpy())
