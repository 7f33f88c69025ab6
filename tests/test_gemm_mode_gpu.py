"""GPU: the opt-in fp16 x3 arithmetic of the feed-forward GEMMs (ppasr_set_gemm_mode, csrc/h3.h) against the torch-CPU
oracle and against the default fp32-MFMA mode.  Every operand is split into two fp16 pieces (22 significant bits), the
products of pieces are exact in fp32 and accumulate in fp32, so the mode is held to the SAME bar as the fp32 kernels:
logits within 1e-3 of the oracle (measured: the fp32 kernels' own 1e-6 class), greedy ids identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.conformer_oracle import ConformerOracle  # noqa: E402
from ppasr_amd import _lib  # noqa: E402
from ppasr_amd.model_utils.conformer.model import ConformerModel  # noqa: E402
from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel  # noqa: E402
from ppasr_amd.utils.synth import conformer_state_dict, squeezeformer_state_dict, synth_features  # noqa: E402


def _model(V, blocks, seed, causal=True):
    sd = conformer_state_dict(vocab_size=V, num_blocks=blocks, seed=seed, perturb_norm=True)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=blocks, cnn_module_kernel=15)
    return sd, ConformerModel(80, V, streaming=causal, encoder_conf=conf, state_dict=sd, device="cuda:0")


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("B,T,lens", [(3, 331, [331, 200, 57]), (9, 403, None), (16, 611, None)])  # conv2 tiles: 32, 96, 128 + 64
def test_f16x3_mode_against_the_oracle_and_the_fp32_mode(B, T, lens):
    V, blocks = 211, 3
    sd, m = _model(V, blocks, 171)
    m.set_row_block(32)  # the mode is built into the 8-wave 32-row kernels: keep small batches off the other block forms
    m.set_ffn_split(0)
    x, la = synth_features(B, T, lens=lens, seed=172 + B)
    m.set_gemm_mode("f32")
    p32, l32 = m.get_encoder_out(x, la, return_logits=True)
    m.set_gemm_mode("f16x3")
    ph, lh = m.get_encoder_out(x, la, return_logits=True)
    l32, lh = l32.cpu().numpy(), lh.cpu().numpy()
    assert np.isfinite(lh).all()
    assert not np.array_equal(l32, lh)  # the mode really ran (its roundings differ from the fp32 MFMAs')
    orc = ConformerOracle(sd, num_blocks=blocks)
    _, lo = orc.get_encoder_out(torch.as_tensor(x), torch.as_tensor(la), return_logits=True)
    lo = lo.numpy()
    e32, eh = _rel(l32, lo), _rel(lh, lo)
    assert e32 < 1e-3 and eh < 1e-3, (e32, eh)
    assert eh < 2e-5, eh  # (measured 1e-6 .. 3e-6: the same class as the fp32 kernels)
    assert np.array_equal(lh.argmax(-1), l32.argmax(-1))
    # back to the default: bit-identical to the first pass
    m.set_gemm_mode("f32")
    _, l32b = m.get_encoder_out(x, la, return_logits=True)
    assert np.array_equal(l32b.cpu().numpy(), l32)


def test_f16x3_mode_on_the_efficient_conformer():
    """Stride layer (stays fp32), grouped-attention layers and the halved depthwise kernel behind the stride layer."""
    from oracle.efficient_conformer_oracle import EfficientConformerOracle
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    from ppasr_amd.utils.synth import efficient_conformer_state_dict
    V, L, stride_idx, groups = 211, 4, 1, (0, 1)
    sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=175, perturb_norm=True, stride_layer_idx=stride_idx,
                                        group_layer_idx=groups)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                cnn_module_norm="layer_norm",
                efficient_conf=dict(stride_layer_idx=[stride_idx], stride=[2], group_layer_idx=list(groups), group_size=3,
                                    stride_kernel=True))
    m = EfficientConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    m.set_row_block(32)
    m.set_ffn_split(0)
    x, la = synth_features(5, 611, lens=[611, 600, 333, 97, 611], seed=176)
    _, l32 = m.get_encoder_out(x, la, return_logits=True)
    m.set_gemm_mode("f16x3")
    _, lh = m.get_encoder_out(x, la, return_logits=True)
    l32, lh = l32.cpu().numpy(), lh.cpu().numpy()
    orc = EfficientConformerOracle(sd, num_blocks=L, stride_layer_idx=stride_idx, group_layer_idx=groups)
    _, lo = orc.get_encoder_out(x, la, return_logits=True)
    assert np.isfinite(lh).all() and not np.array_equal(lh, l32)
    assert _rel(lh, lo.numpy()) < 2e-5 and _rel(l32, lo.numpy()) < 2e-5
    assert np.array_equal(lh.argmax(-1), l32.argmax(-1))


def test_f16x3_mode_at_the_baseline_shape_with_ragged_lengths():
    """BASELINE configs[1]'s shape (32 x 1000 frames, 12 blocks, V = 4233): full launches take the 32-row kernels by
    themselves; ragged lengths + skip_padding exercise the padded rows and the active-block lists."""
    V, blocks = 4233, 12
    sd, m = _model(V, blocks, 1234)
    rng = np.random.default_rng(5)
    lens = [1000] + [int(v) for v in rng.integers(300, 1001, size=31)]
    x, la = synth_features(32, 1000, lens=lens, seed=20240 + 7)
    m.set_gemm_mode("f32")
    _, l32 = m.get_encoder_out(x, la, return_logits=True)
    m.set_gemm_mode("f16x3")
    _, lh = m.get_encoder_out(x, la, return_logits=True)
    m.set_skip_padding(True)
    _, lhs = m.get_encoder_out(x, la, return_logits=True)
    m.set_skip_padding(False)
    l32, lh, lhs = l32.cpu().numpy(), lh.cpu().numpy(), lhs.cpu().numpy()
    assert np.isfinite(lh).all() and not np.array_equal(lh, l32)
    assert _rel(lh, l32) < 2e-5
    tp = lh.shape[1]
    tv = [min(tp, (int(n) + 3) // 4) for n in la]  # frame t is valid iff 4 t < len (subsampling.py:115)
    for b in range(32):
        assert np.array_equal(lh[b, :tv[b]].argmax(-1), l32[b, :tv[b]].argmax(-1)), b
        assert np.array_equal(lhs[b, :tv[b]], lh[b, :tv[b]]), b  # ragged mode: valid rows bit-identical within the mode


def test_f16x3_mode_on_the_squeezeformer():
    """Squeezeformer handles: conv2 of the front end and both feed-forward modules of the 32-row layer kernels."""
    from oracle.squeezeformer_oracle import SqueezeformerOracle
    V = 97
    sq_sd = squeezeformer_state_dict(vocab_size=V, num_blocks=2, seed=92, perturb_norm=True)
    sq_conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=2, reduce_idx=None, recover_idx=None,
                   feed_forward_expansion_factor=8, cnn_module_kernel=31)
    sm = SqueezeformerModel(80, V, streaming=True, encoder_conf=sq_conf, state_dict=sq_sd, device="cuda:0")
    sm.set_row_block(32)
    sm.set_ffn_split(0)
    x, la = synth_features(6, 611, lens=[611, 600, 333, 97, 611, 13], seed=94)
    _, l32 = sm.get_encoder_out(x, la, return_logits=True)
    sm.set_gemm_mode("f16x3")
    _, lh = sm.get_encoder_out(x, la, return_logits=True)
    orc = SqueezeformerOracle(sq_sd, num_blocks=2, cnn_module_kernel=31, reduce_idx=None, recover_idx=None)
    _, lo = orc.get_encoder_out(x, la, return_logits=True)
    assert _rel(lh.cpu().numpy(), lo.numpy()) < 2e-5
    sm.set_skip_padding(True)
    _, lhs = sm.get_encoder_out(x, la, return_logits=True)
    l32, lh, lhs = l32.cpu().numpy(), lh.cpu().numpy(), lhs.cpu().numpy()
    assert np.isfinite(lh).all() and not np.array_equal(lh, l32)
    assert _rel(lh, l32) < 2e-5
    assert np.array_equal(lh.argmax(-1), l32.argmax(-1))
    for b, n in enumerate(la):
        tv = min(lh.shape[1], (int(n) + 3) // 4)
        assert np.array_equal(lhs[b, :tv], lh[b, :tv]), b


def test_f16x3_mode_is_refused_where_it_is_not_built():
    # the general layer route (width 512) has neither the fused layer kernels nor their front end
    sd = conformer_state_dict(vocab_size=97, num_blocks=1, seed=4, output_size=512, attention_heads=8)
    conf = dict(output_size=512, attention_heads=8, linear_units=2048, num_blocks=1, cnn_module_kernel=15)
    wide = ConformerModel(80, 97, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    assert wide.lib.ppasr_set_gemm_mode(wide._h, _lib.PPASR_GEMM_F16X3) != 0
    assert wide.lib.ppasr_set_gemm_mode(wide._h, _lib.PPASR_GEMM_F32) == 0
    sd, m = _model(97, 1, 3)
    assert m.lib.ppasr_set_gemm_mode(m._h, 7) != 0


# ---- range guard (csrc/h3.h, ppasr_set_gemm_guard / ppasr_gemm_guard_stats) ------------------------------------------
def test_f16x3_range_guard_falls_back_to_fp32_and_counts():
    """A GEMM input beyond the fp16 pieces' range (features x 1e4: ReLU(conv1) in front of conv2 reaches ~1e5 > 4 094) -- guard
    on: the call returns the fp32 mode's result BIT FOR BIT (it was re-run on the fp32 kernels) and the fallback is counted;
    guard off: finite (saturated) output, events counted, no fallback.  In-range inputs: no event, no fallback."""
    V, blocks = 157, 2
    sd, m = _model(V, blocks, 301)
    m.set_row_block(32)
    m.set_ffn_split(0)
    x, la = synth_features(3, 203, lens=[203, 150, 64], seed=302)
    big = (x * 1e4).astype(np.float32)
    m.set_gemm_mode("f32")
    _, ref_big = m.get_encoder_out(big, la, return_logits=True)
    _, ref = m.get_encoder_out(x, la, return_logits=True)
    ref_big, ref = ref_big.cpu().numpy(), ref.cpu().numpy()
    assert np.isfinite(ref_big).all()
    m.set_gemm_mode("f16x3")
    assert m.gemm_coverage() == {"layers", "front", "head"}
    assert m.gemm_guard_stats() == (0, 0)
    # in range: the mode runs, nothing is counted
    _, lh = m.get_encoder_out(x, la, return_logits=True)
    lh = lh.cpu().numpy()
    assert not np.array_equal(lh, ref) and _rel(lh, ref) < 2e-5
    assert m.gemm_guard_stats() == (0, 0)
    # out of range, guard on (default)
    _, lb = m.get_encoder_out(big, la, return_logits=True)
    lb = lb.cpu().numpy()
    f, e = m.gemm_guard_stats()
    print(f"guard on: fallbacks {f}, events {e}")
    assert f == 1 and e > 0
    assert np.array_equal(lb, ref_big)
    tokens_g, n_g, _ = m.encode_greedy(big, la)
    assert m.gemm_guard_stats()[0] == 2
    m.set_gemm_mode("f32")
    tokens_r, n_r, _ = m.encode_greedy(big, la)
    assert torch.equal(tokens_g, tokens_r) and torch.equal(n_g, n_r)
    # out of range, guard off: saturated, finite, counted -- and not the fp32 result
    m.set_gemm_mode("f16x3")
    m.set_gemm_guard(False)
    _, ls = m.get_encoder_out(big, la, return_logits=True)
    ls = ls.cpu().numpy()
    f2, e2 = m.gemm_guard_stats()
    print(f"guard off: fallbacks {f2}, events {e2}")
    assert np.isfinite(ls).all()
    assert f2 == 2 and e2 > e
    assert not np.array_equal(ls, ref_big)
    # guard back on, in-range input: the mode's own result again, no new fallback
    m.set_gemm_guard(True)
    _, lh2 = m.get_encoder_out(x, la, return_logits=True)
    assert np.array_equal(lh2.cpu().numpy(), lh)
    assert m.gemm_guard_stats()[0] == 2


def test_f16x3_range_guard_squeezeformer():
    """The Squeezeformer kernels live in another translation unit (their own event counter): swish(hidden) beyond the range
    through a scaled feed-forward weight... is not reachable without touching the checkpoint, so the feature route again
    (the front end is shared) plus an in-range pass that must count nothing."""
    V, L = 101, 3
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=311, perturb_norm=True, streaming=True)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=1, recover_idx=2,
                feed_forward_expansion_factor=8, cnn_module_kernel=31)
    m = SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    m.set_row_block(32)
    x, la = synth_features(3, 203, lens=[203, 150, 64], seed=312)
    m.set_gemm_mode("f32")
    _, ref = m.get_encoder_out((x * 1e4).astype(np.float32), la, return_logits=True)
    m.set_gemm_mode("f16x3")
    _, a = m.get_encoder_out(x, la, return_logits=True)
    assert m.gemm_guard_stats() == (0, 0) and np.isfinite(a.cpu().numpy()).all()
    _, b = m.get_encoder_out((x * 1e4).astype(np.float32), la, return_logits=True)
    assert m.gemm_guard_stats()[0] == 1
    assert torch.equal(b, ref)


def test_f16x3_mode_refuses_weights_beyond_the_fp16_range():
    """|w| >= 255.9 does not fit the 2^8-scaled pieces: ppasr_set_gemm_mode fails with PPASR_EUNSUPPORTED, the handle stays in
    the default mode and keeps working."""
    V, blocks = 157, 2
    sd = conformer_state_dict(vocab_size=V, num_blocks=blocks, seed=303, perturb_norm=True)
    k = "encoder.encoders.1.feed_forward.w_1.weight"
    sd[k] = sd[k].copy()
    sd[k][3, 5] = 300.0
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=blocks, cnn_module_kernel=15)
    m = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    x, la = synth_features(2, 131, seed=304)
    _, before = m.get_encoder_out(x, la, return_logits=True)
    with pytest.raises(_lib.PPASRHipError) as ei:
        m.set_gemm_mode("f16x3")
    assert ei.value.status == _lib.PPASR_EUNSUPPORTED
    assert m.gemm_coverage() == set()
    _, after = m.get_encoder_out(x, la, return_logits=True)
    assert torch.equal(before, after)


def test_f16x3_coverage_reports_uncovered_layer_kernels():
    """cnn_module_kernel 31 on a Conformer: the layer kernels have no fp16 x3 form (7 / 15 only) -- the mode is accepted for
    the front end and the head, and ppasr_gemm_coverage says so."""
    V, blocks = 157, 1
    sd = conformer_state_dict(vocab_size=V, num_blocks=blocks, seed=305, cnn_module_kernel=31)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=blocks, cnn_module_kernel=31)
    m = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    m.set_gemm_mode("f16x3")
    assert m.gemm_coverage() == {"front", "head"}
    m.set_gemm_mode("f32")
    assert m.gemm_coverage() == set()


@pytest.mark.parametrize("family", ["conformer", "efficient"])
@pytest.mark.parametrize("lens", [[333, 280, 120], [67]])
def test_f16x3_mode_on_the_split_route(family, lens):
    """Under-filled launches (a few row blocks: small batches, single utterances) take the split route; in the mode its
    kernels' GEMM units run on the fp16 x3 route too (k_ffn_part / k_ln_qkv / k_out_glu / k_pw1_glu_cols / k_conv_pre <.., true>):
    logits against the oracle and the default mode, per-frame argmax unchanged."""
    from oracle.conformer_oracle import ConformerOracle
    from oracle.efficient_conformer_oracle import EfficientConformerOracle
    from ppasr_amd._lib import kernel_profile
    V, L = 157, 4
    if family == "conformer":
        from ppasr_amd.model_utils.conformer.model import ConformerModel
        sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=901)
        conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
        model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
        oracle = ConformerOracle(sd, num_blocks=L)
    else:
        from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
        from ppasr_amd.utils.synth import efficient_conformer_state_dict
        sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=902, perturb_norm=True, stride_layer_idx=1,
                                            group_layer_idx=(0, 1))
        conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                    cnn_module_norm="layer_norm",
                    efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3, stride_kernel=True))
        model = EfficientConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
        oracle = EfficientConformerOracle(sd, num_blocks=L, stride_layer_idx=1, group_layer_idx=(0, 1))
    x, la = synth_features(len(lens), max(lens), lens=lens, seed=903)
    _, base = model.get_encoder_out(x, la, return_logits=True)
    base = base.cpu().numpy()
    model.set_gemm_mode("f16x3")
    with kernel_profile() as kp:
        _, got = model.get_encoder_out(x, la, return_logits=True)
        torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert any(k.startswith("k_ffn_part<true>") for k in kp.kernels), sorted(kp.kernels)
    _, ref = oracle.get_encoder_out(x, la, return_logits=True)
    ref = ref.numpy()
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() / scale < 2e-5
    assert np.abs(got - base).max() / scale < 2e-5
    for b, n in enumerate(lens):
        tp = ref.shape[1] if family == "conformer" else got.shape[1]
        assert np.array_equal(got[b].argmax(-1), base[b].argmax(-1)), b


def test_f16x3_mode_on_stream_handles_and_session_groups():
    """A stream handle / a session group of a model in the mode: the chunk's split-route kernels run their units on the
    fp16 x3 route; chunk by chunk against the default mode's stream (pinned to the reference source in
    tests/test_ref_pin_gpu.py, which runs in the mode as well)."""
    from ppasr_amd._lib import kernel_profile
    from ppasr_amd.model_utils.conformer.model import ConformerModel, ConformerStreamGroup
    V, L = 157, 3
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=911)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
    model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    x, _ = synth_features(2, 64 * 3 + 67, seed=912)
    wins = [(c, min(c + 67, x.shape[1])) for c in range(0, x.shape[1] - 7 + 1, 64)]

    def run():
        s = model.new_stream()
        single = [s.encode_chunk(x[:1, a:b], -16).cpu().numpy() for a, b in wins]
        grp = ConformerStreamGroup(model, 2, max_frames=16 * 8)
        fa = [grp.encode_chunks([0, 1], torch.from_numpy(x[:, a:b]).cuda())[0].cpu().numpy() for a, b in wins if b - a == 67]
        return single, fa

    base_single, base_fa = run()
    model.set_gemm_mode("f16x3")
    with kernel_profile() as kp:
        got_single, got_fa = run()
        torch.cuda.synchronize()
    assert any(k.startswith("k_ffn_part<true>") for k in kp.kernels) and any(k.startswith("k_ln_qkv<true>") for k in kp.kernels)
    for a, b in zip(got_single, base_single):
        assert a.shape == b.shape and np.abs(a - b).max() < 1e-5  # (probabilities)
        assert np.array_equal(a.argmax(-1), b.argmax(-1))
    for a, b in zip(got_fa, base_fa):
        assert np.array_equal(a, b)  # per-frame argmax of both sessions


def test_f16x3_mode_on_the_squeezeformer_split_route_and_stream():
    """Squeezeformer in the mode on under-filled launches and stream handles: the feed-forward slices (k_ffn_part<true>) on the
    fp16 x3 route, the rest of the split route in fp32; against the oracle (batched) and the default mode's stream (chunks)."""
    from oracle.squeezeformer_oracle import SqueezeformerOracle
    from ppasr_amd._lib import kernel_profile
    V, L = 97, 3
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=921, perturb_norm=True)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=1, recover_idx=2,
                feed_forward_expansion_factor=8, cnn_module_kernel=31)
    sm = SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    x, la = synth_features(2, 331, lens=[331, 200], seed=922)
    xs, _ = synth_features(1, 64 * 2 + 67, seed=923)
    wins = [(c, min(c + 67, xs.shape[1])) for c in range(0, xs.shape[1] - 7 + 1, 64)]

    def stream_out():
        s = sm.new_stream()
        return [s.encode_chunk(xs[:, a:b], -16).cpu().numpy() for a, b in wins]

    base_chunks = stream_out()
    sm.set_gemm_mode("f16x3")
    with kernel_profile() as kp:
        _, lh = sm.get_encoder_out(x, la, return_logits=True)
        got_chunks = stream_out()
        torch.cuda.synchronize()
    assert any(k.startswith("k_ffn_part<true>") for k in kp.kernels), sorted(kp.kernels)
    _, lo = SqueezeformerOracle(sd, num_blocks=L, cnn_module_kernel=31, reduce_idx=1, recover_idx=2).get_encoder_out(
        x, la, return_logits=True)
    assert _rel(lh.cpu().numpy(), lo.numpy()) < 2e-5
    for a, b in zip(got_chunks, base_chunks):
        assert a.shape == b.shape and np.abs(a - b).max() < 1e-5
        assert np.array_equal(a.argmax(-1), b.argmax(-1))
