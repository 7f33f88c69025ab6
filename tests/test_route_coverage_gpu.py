"""Routes that no other GPU test reaches (found by tools/kernel_coverage.py: kernels compiled into the library that the
suite never launched): the FUSED layer kernels behind the stream handles (`ppasr_set_ffn_split(h, 0)`: k_conv_ffn<KS, STREAM>,
the 8-wave k_sq_tail<KS>), Squeezeformer models with `cnn_module_kernel: 15` on every block form and in the fp16 x3 mode, and
the single-utterance GRU step kernel."""
import numpy as np
import pytest
import torch

from ppasr_amd.utils.synth import (conformer_state_dict, deepspeech2_state_dict, efficient_conformer_state_dict,
                                   squeezeformer_state_dict, synth_features)

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _windows(n_frames, window=67, stride=64):
    return [(cur, min(cur + window, n_frames)) for cur in range(0, n_frames - 7 + 1, stride)]


def _model(family, ks, V, L):
    if family == "conformer":
        from ppasr_amd.model_utils.conformer.model import ConformerModel
        sd = conformer_state_dict(vocab_size=V, num_blocks=L, cnn_module_kernel=ks, seed=500 + ks)
        conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=ks)
        return ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    if family == "efficient":
        from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
        sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=510, perturb_norm=True, stride_layer_idx=1,
                                            group_layer_idx=(0, 1))
        conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                    cnn_module_norm="layer_norm",
                    efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3, stride_kernel=True))
        return EfficientConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, cnn_module_kernel=ks, seed=520 + ks, perturb_norm=True)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=1, recover_idx=3,
                feed_forward_expansion_factor=8, cnn_module_kernel=ks)
    return SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")


@pytest.mark.parametrize("family,ks", [("conformer", 15), ("conformer", 31), ("efficient", 15), ("squeezeformer", 31),
                                       ("squeezeformer", 15)])
def test_stream_chunks_on_the_fused_route_equal_the_split_route(family, ks):
    """A stream handle runs its one row block on the split route by default (the feed-forward modules over 8 workgroups); with
    ppasr_set_ffn_split(h, 0) the same chunks go through the fused layer kernels with the conv history
    (k_conv_ffn<KS, true, false>: 15 / 31 taps, 7 behind the Efficient-Conformer's stride layer; Squeezeformer:
    k_sq_mid + the 8-wave k_sq_tail<KS>).  The split route is pinned to the reference source chunk by chunk
    (tests/test_ref_pin_gpu.py); the two routes differ only in the order of the sum over hidden chunks."""
    V, L = 150, 4
    model = _model(family, ks, V, L)
    x, _ = synth_features(1, 64 * 3 + 67, seed=530)
    outs = {}
    for route in (-1, 0):
        model.set_ffn_split(route)
        stream = model.new_stream()
        outs[route] = [stream.encode_chunk(x[:, a:b], -16).cpu().numpy() for a, b in _windows(x.shape[1])]
        torch.cuda.synchronize()
    model.set_ffn_split(-1)
    assert len(outs[-1]) == len(outs[0]) == 4
    for i, (s, f) in enumerate(zip(outs[-1], outs[0])):
        assert s.shape == f.shape and np.isfinite(f).all()
        assert _rel(f, s) < 2e-5, (family, ks, i, _rel(f, s))


@pytest.mark.parametrize("mode", ["rows32", "rows16", "rows32_w16", "f16x3"])
def test_squeezeformer_with_15_tap_conv_modules(mode):
    """`cnn_module_kernel: 15` (the Squeezeformer constructor takes any odd kernel, squeezeformer/encoder.py:40; the YAML ships
    31): the 15-tap instantiations of k_sq_tail on every block form and in the fp16 x3 mode, against the oracle."""
    from oracle.squeezeformer_oracle import SqueezeformerOracle
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    V, L = 131, 4
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, cnn_module_kernel=15, seed=540, perturb_norm=True)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=1, recover_idx=3,
                feed_forward_expansion_factor=8, cnn_module_kernel=15)
    model = SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    model.set_ffn_split(0)  # (the batch is a few row blocks: keep it on the fused layer kernels)
    if mode == "f16x3":
        model.set_gemm_mode("f16x3")
        model.set_row_block(32)
    else:
        model.set_row_block({"rows32": 32, "rows16": 16, "rows32_w16": 1032}[mode])
    x, la = synth_features(3, 403, lens=[403, 350, 167], seed=541)
    _, logits = model.get_encoder_out(x, la, return_logits=True)
    _, ref = SqueezeformerOracle(sd, num_blocks=L, cnn_module_kernel=15, reduce_idx=1, recover_idx=3).get_encoder_out(
        x, la, return_logits=True)
    torch.cuda.synchronize()
    e = _rel(logits.cpu().numpy(), ref.numpy())
    print(f"squeezeformer, 15 taps, {mode}: logits {e:.2e}")
    assert e < 1e-3


@pytest.mark.parametrize("ks", [31, 15])
def test_squeezeformer_utterances_of_three_frames(ks):
    """T = 15 feature frames = 3 encoder frames: fewer than the 4 consecutive rows a wave of the register depthwise conv
    walks, so the layer tail takes the LDS-staged 8-wave form (k_sq_tail<KS, false>)."""
    from oracle.squeezeformer_oracle import SqueezeformerOracle
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    V, L = 131, 3
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, cnn_module_kernel=ks, seed=560 + ks, perturb_norm=True)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=None, recover_idx=None,
                feed_forward_expansion_factor=8, cnn_module_kernel=ks)
    model = SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    model.set_ffn_split(0)
    x, la = synth_features(5, 15, lens=[15, 15, 13, 15, 11], seed=561)
    _, logits = model.get_encoder_out(x, la, return_logits=True)
    _, ref = SqueezeformerOracle(sd, num_blocks=L, cnn_module_kernel=ks, reduce_idx=None, recover_idx=None).get_encoder_out(
        x, la, return_logits=True)
    torch.cuda.synchronize()
    assert tuple(logits.shape) == tuple(ref.shape) and logits.shape[1] == 3
    assert _rel(logits.cpu().numpy(), ref.numpy()) < 1e-3


def test_gru_single_utterance_step_kernel():
    """One utterance through a GRU stack (use_gru: True, B = 1): the per-step VALU kernel k_gru_step (batches take the
    matrix-core step or the wavefront)."""
    from oracle.deepspeech2_oracle import DeepSpeech2Oracle
    from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
    V, L, H = 97, 2, 1024
    for streaming in (True, False):
        sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, rnn_size=H, streaming=streaming, seed=550, perturb_norm=True,
                                    use_gru=True)
        x, lens = synth_features(1, 131, lens=[131], seed=551)
        model = DeepSpeech2Model(80, V, streaming=streaming, encoder_conf=dict(num_rnn_layers=L, rnn_size=H, use_gru=True),
                                 state_dict=sd, device="cuda:0")
        probs, out_lens, fh, _fc = model.get_encoder_out_chunk(x, lens)
        torch.cuda.synchronize()
        rp, rl, rh, _rc = DeepSpeech2Oracle(sd, L, H, streaming, use_gru=True).forward(x, lens)
        assert out_lens.cpu().tolist() == rl.tolist()
        assert _rel(probs.cpu().numpy(), rp.numpy()) < 1e-3
        assert _rel(fh.cpu().numpy(), rh.numpy()) < 1e-3


def test_probabilities_of_a_vocabulary_beyond_5120_characters():
    """The softmax of the probability rows keeps a row in registers up to 5 120 columns (k_softmax_row_wg<8 / 20>); larger
    vocabularies take the form that re-reads the row (k_softmax_row_wg<0>): a one-block Conformer with 5 500 characters
    against the oracle, rows summing to one."""
    from oracle.conformer_oracle import ConformerOracle
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    V, L = 5500, 1
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=77)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
    model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    x, lens = synth_features(2, 131, lens=[131, 90], seed=78)
    got = model.get_encoder_out(x, lens).cpu().numpy()
    want = ConformerOracle(sd, num_blocks=L).get_encoder_out(x, lens).numpy()
    assert got.shape == want.shape
    assert _rel(got, want) < 1e-3
    assert np.abs(got.sum(-1) - 1.0).max() < 1e-5
