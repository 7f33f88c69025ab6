"""GPU: the layer kernels on the block forms of csrc/rbt.h (phases_t.h) -- 32-row blocks on 8 waves are bit-identical to
the round-3 kernels they replace (same sums, same order); 16-row blocks and 32 rows on 16 waves (both
v_mfma_f32_16x16x4_f32) agree to rounding, hold the oracle tolerance on their own, and the route choice
(ppasr_set_row_block, ppasr_set_lengths_hint) never changes which rows are computed."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle.squeezeformer_oracle import SqueezeformerOracle
from ppasr_amd.utils.synth import squeezeformer_state_dict, synth_features

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V, L, RED, REC = 300, 5, 2, 4
W16 = 1032  # PPASR_ROW_BLOCK_32_W16
# 16 utterances of 152 frames: 76 blocks of 32 rows on the full-rate layers, 38 on the reduced ones -- both inside the
# window (33 .. 128 blocks) where the grid-size rule picks the 16-row kernels
LENS = [611, 600, 333, 97, 611, 13, 250, 480, 611, 420, 77, 590, 611, 305, 150, 555]


def _model(streaming=True, kernel=31, norm="layer_norm"):
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=71, perturb_norm=True, cnn_module_kernel=kernel,
                                  streaming=streaming, cnn_norm_type=norm)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=RED, recover_idx=REC,
                feed_forward_expansion_factor=8, cnn_module_kernel=kernel, cnn_norm_type=norm)
    return SqueezeformerModel(80, V, streaming=streaming, encoder_conf=conf, state_dict=sd, device="cuda:0"), sd


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _run(model, x, lens, rows):
    model.set_row_block(rows)
    try:
        probs, logits = model.get_encoder_out(x, lens, return_logits=True)
        torch.cuda.synchronize()
        return probs.cpu().numpy(), logits.cpu().numpy()
    finally:
        model.set_row_block(-1)


@pytest.mark.parametrize("streaming,kernel,norm", [(True, 31, "layer_norm"), (False, 31, "layer_norm"), (True, 15, "batch_norm")])
def test_16_row_blocks_match_32_row_blocks_and_the_oracle(streaming, kernel, norm):
    model, sd = _model(streaming, kernel, norm)
    x, lens = synth_features(len(LENS), 611, lens=LENS, seed=72)
    p32, l32 = _run(model, x, lens, 32)
    p16, l16 = _run(model, x, lens, 16)
    pw, lw = _run(model, x, lens, W16)
    pa, la = _run(model, x, lens, -1)   # the grid-size rule picks the 16-row kernels for every layer of this batch
    assert np.array_equal(la, l16)
    e, ew = _rel(l16, l32), _rel(lw, l32)
    print("16 vs 32 rows: max rel logit diff", e, " 32 rows on 16 waves vs 8:", ew)
    assert e < 2e-5 and ew < 2e-5
    oracle = SqueezeformerOracle(sd, num_blocks=L, cnn_module_kernel=kernel, reduce_idx=RED, recover_idx=REC, causal=streaming)
    with torch.no_grad():
        enc, _ = oracle.encoder_forward(x, lens)[:2]
        ref = oracle.ctc_logits(enc).numpy()
    e16, e32, ew16 = _rel(l16, ref), _rel(l32, ref), _rel(lw, ref)
    print("vs oracle: 16-row", e16, "32-row", e32, "32 rows on 16 waves", ew16)
    assert e16 < 1e-3 and e32 < 1e-3 and ew16 < 1e-3
    near_tie = (np.sort(p32, -1)[..., -1] - np.sort(p32, -1)[..., -2]).min() < 1e-5
    assert np.array_equal(p16.argmax(-1), p32.argmax(-1)) or near_tie
    assert np.array_equal(pw.argmax(-1), p32.argmax(-1)) or near_tie


def test_16_wave_kernels_on_a_full_launch_with_a_partial_last_block():
    """32 utterances x 151 frames = 151 blocks of 32 rows (the last one 16 rows): 32 rows on 16 waves (an option,
    PPASR_ROW_BLOCK_32_W16) against the 8-wave kernels; the grid-size rule mixes 32-row blocks at the full rate with
    16-row blocks at the reduced one (76 blocks)."""
    model, _ = _model()
    lens_l = (LENS + LENS[::-1])
    lens_l = [min(v, 607) for v in lens_l]
    x, lens = synth_features(len(lens_l), 607, lens=lens_l, seed=75)
    pa, la = _run(model, x, lens, -1)
    pw, lw = _run(model, x, lens, W16)
    p32, l32 = _run(model, x, lens, 32)
    fl = model.valid_out_frames(lens, 607).cpu().numpy()
    # the auto route mixes forms per layer; both it and the forced form stay within rounding of the 8-wave kernels
    for u in range(len(lens_l)):
        n = int(fl[u])
        assert _rel(la[u, :n], l32[u, :n]) < 2e-5 and _rel(lw[u, :n], l32[u, :n]) < 2e-5
    assert not np.array_equal(lw, l32)   # (a different summation order: the 16-wave kernels did run)


def test_ragged_mode_with_hint_computes_the_same_valid_rows():
    """skip_padding + lengths hint: the hint only picks the kernel variant; valid rows equal the unhinted ragged run of the
    same variant bit for bit, and the 16-row variant stays within rounding of the 32-row one."""
    model, _ = _model()
    # 24 utterances: with the hint the computed rows are ~78 blocks of 32 at the full rate and ~41 at the reduced one --
    # inside the window where the rule picks 16-row blocks for every layer (<= 32 blocks would take the split route)
    LENS = [611, 600, 333, 197, 611, 213, 250, 480, 611, 420, 177, 590, 611, 305, 250, 555, 380, 611, 444, 290, 505, 160, 611, 350]
    x, lens = synth_features(len(LENS), 611, lens=LENS, seed=73)
    fl = model.valid_out_frames(lens, 611).cpu().numpy()
    model.set_skip_padding(True)
    try:
        model.set_row_block(16)
        a = model.get_encoder_out(x, lens).cpu().numpy()
        model.set_row_block(-1)
        model.set_lengths_hint(LENS)
        b = model.get_encoder_out(x, lens).cpu().numpy()   # hint: few active rows -> 16-row kernels
        model.set_lengths_hint(None)
        model.set_row_block(32)
        c = model.get_encoder_out(x, lens).cpu().numpy()
        model.set_row_block(W16)
        d = model.get_encoder_out(x, lens).cpu().numpy()
    finally:
        model.set_row_block(-1)
        model.set_lengths_hint(None)
        model.set_skip_padding(False)
    for u in range(len(LENS)):
        n = int(fl[u])
        assert np.array_equal(a[u, :n], b[u, :n])
        assert _rel(a[u, :n], c[u, :n]) < 2e-5 and _rel(d[u, :n], c[u, :n]) < 2e-5
        assert not a[u, n:].any() and not b[u, n:].any() and not c[u, n:].any() and not d[u, n:].any()


@pytest.mark.parametrize("family", ["conformer", "conformer_noncausal_k31", "efficient"])
def test_conformer_family_16_row_blocks(family):
    """The Conformer / Efficient-Conformer layer kernels on 16-row blocks (csrc/conformer_kernels_t.hip: k_ffn_qkv_t,
    k_out_glu_t, k_conv_ffn_t with the next layer's S1 fused in, the stand-alone attention between them) against the fused
    32-row route (ppasr_set_ffn_split(0)) and the oracle; the grid-size rule picks them by itself for 33 .. 128 row blocks."""
    from oracle.conformer_oracle import ConformerOracle
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    from ppasr_amd.utils.synth import conformer_state_dict, efficient_conformer_state_dict
    V, L = 211, 4
    lens = [611, 600, 333, 97, 611, 13, 250, 480, 611, 420, 77, 590]   # 12 x 152 frames = 57 blocks of 32 rows
    if family == "efficient":
        from oracle.efficient_conformer_oracle import EfficientConformerOracle
        from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
        sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=81, perturb_norm=True, stride_layer_idx=1,
                                            group_layer_idx=(0, 1))
        conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                    cnn_module_norm="layer_norm",
                    efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3, stride_kernel=True))
        model = EfficientConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
        oracle = EfficientConformerOracle(sd, num_blocks=L, stride_layer_idx=1, group_layer_idx=(0, 1))
        lens = lens + lens   # 24 utterances: 114 blocks at the full rate, 57 behind the stride layer
    else:
        k, streaming = (31, False) if family.endswith("k31") else (15, True)
        sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=82, perturb_norm=True, cnn_module_kernel=k)
        conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=k)
        model = ConformerModel(80, V, streaming=streaming, encoder_conf=conf, state_dict=sd, device="cuda:0")
        oracle = ConformerOracle(sd, num_blocks=L, cnn_module_kernel=k, causal=streaming)
    x, la = synth_features(len(lens), 611, lens=lens, seed=83)
    model.set_ffn_split(0)
    p32, l32 = _run(model, x, la, 32)          # the fused 32-row kernels
    model.set_ffn_split(-1)
    p16, l16 = _run(model, x, la, 16)
    pa, la_ = _run(model, x, la, -1)           # 57 / 114 blocks: the rule picks the 16-row kernels
    assert np.array_equal(la_, l16)
    # 32 rows on 16 waves inside the fused route (ppasr_set_ffn_split(0)): k_ffn_qkv_t / k_conv_ffn_t<kW16> write the values
    # in the fragment order the fused attention kernel reads; grouped-attention layers take k_out_glu_t<kW16>
    model.set_ffn_split(0)
    pw, lw = _run(model, x, la, W16)
    model.set_ffn_split(-1)
    e, ew = _rel(l16, l32), _rel(lw, l32)
    ref = oracle.get_encoder_out(x, la, return_logits=True)[1].numpy()
    e16, e32, ew16 = _rel(l16, ref), _rel(l32, ref), _rel(lw, ref)
    print(family, "16 vs 32 rows", e, "16 waves vs 8", ew, "vs oracle: 16-row", e16, "32-row", e32, "16 waves", ew16)
    assert e < 2e-5 and ew < 2e-5 and e16 < 1e-3 and e32 < 1e-3 and ew16 < 1e-3
    assert not np.array_equal(lw, l32)


def test_conformer_16_wave_kernels_around_the_fused_attention():
    """40 x 151 frames = 189 blocks (the last one partial), ragged lengths, skip_padding on and off: the fused attention
    kernel between 16-wave layer kernels (which write its fragment-ordered values); same valid rows as the default route
    (8 waves) to rounding."""
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    from ppasr_amd.utils.synth import conformer_state_dict
    V, L = 211, 3
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=84, perturb_norm=True)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
    model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    rng = np.random.default_rng(85)
    lens_l = [607] + [int(v) for v in rng.integers(40, 608, size=39)]
    x, lens = synth_features(len(lens_l), 607, lens=lens_l, seed=86)
    fl = model.valid_out_frames(lens, 607).cpu().numpy()
    for skip in (False, True):
        model.set_skip_padding(skip)
        try:
            pa, la = _run(model, x, lens, -1)
            pw, lw = _run(model, x, lens, W16)
            p32, l32 = _run(model, x, lens, 32)
        finally:
            model.set_skip_padding(False)
        assert np.array_equal(la, l32)          # 189 blocks: the rule picks the 8-wave 32-row kernels
        assert not np.array_equal(lw, l32)
        for u in range(len(lens_l)):
            n = int(fl[u])
            assert _rel(lw[u, :n], l32[u, :n]) < 2e-5
            if skip:
                assert not lw[u, n:].any()
