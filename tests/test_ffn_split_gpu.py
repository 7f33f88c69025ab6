"""The split route for under-filled launches (ppasr_set_ffn_split): the layer tail cut at its feed-forward modules, each
module's hidden dimension split over S workgroups per row block.  Every mode must stay within the logit tolerance of the
oracle and give the oracle's greedy tokens; mode 0 (always the fused kernels) keeps the fused route covered at the small
shapes of this test suite, where the default (-1) picks the split route."""
import numpy as np
import pytest
import torch

from oracle.ctc_decoders_oracle import greedy_tokens
from ppasr_amd.utils.synth import (conformer_state_dict, efficient_conformer_state_dict, squeezeformer_state_dict,
                                   synth_features)

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _conformer(streaming):
    from oracle.conformer_oracle import ConformerOracle
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    V, L = 97, 3
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=21, perturb_norm=True)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
    return (ConformerModel(80, V, streaming=streaming, encoder_conf=conf, state_dict=sd, device="cuda:0"),
            ConformerOracle(sd, num_blocks=L, causal=streaming) if not streaming else ConformerOracle(sd, num_blocks=L))


def _efficient():
    from oracle.efficient_conformer_oracle import EfficientConformerOracle
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    V, L = 113, 4
    sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=22, perturb_norm=True, stride_layer_idx=1,
                                        group_layer_idx=(0, 1))
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                cnn_module_norm="layer_norm",
                efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3,
                                    stride_kernel=True))
    return (EfficientConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0"),
            EfficientConformerOracle(sd, num_blocks=L, stride_layer_idx=1, group_layer_idx=(0, 1)))


def _squeezeformer():
    from oracle.squeezeformer_oracle import SqueezeformerOracle
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    V, L = 131, 4
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=23, perturb_norm=True)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=1, recover_idx=3,
                feed_forward_expansion_factor=8, cnn_module_kernel=31)
    return (SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0"),
            SqueezeformerOracle(sd, num_blocks=L, reduce_idx=1, recover_idx=3))


@pytest.mark.parametrize("family", ["conformer", "efficient", "squeezeformer"])
@pytest.mark.parametrize("lens", [[333, 280, 120], [67]])
def test_every_split_mode_matches_the_oracle(family, lens):
    model, oracle = {"conformer": lambda: _conformer(True), "efficient": _efficient, "squeezeformer": _squeezeformer}[family]()
    x, la = synth_features(len(lens), max(lens), lens=lens, seed=sum(lens))
    ref_probs, ref_logits = oracle.get_encoder_out(x, la, return_logits=True)
    outs = {}
    for mode in (0, 2, 4, 8, -1):
        model.set_ffn_split(mode)
        probs, logits = model.get_encoder_out(x, la, return_logits=True)
        tokens, n, _ = model.encode_greedy(x, la)
        torch.cuda.synchronize()
        assert _rel(logits.cpu().numpy(), ref_logits.numpy()) < TOL, mode
        for b in range(len(lens)):
            ids, _, _ = greedy_tokens(ref_probs[b].numpy())
            assert np.array_equal(ids, tokens[b, :int(n[b])].cpu().numpy()), (mode, b)
        outs[mode] = logits
    model.set_ffn_split(-1)
    # the routes differ only in the order of the final sum over hidden chunks
    assert _rel(outs[8].cpu().numpy(), outs[0].cpu().numpy()) < 1e-5
    assert torch.equal(outs[-1], outs[8])  # <= 128 row blocks and n_chunks = 8: the default picks 8 slices here


def test_split_mode_with_ragged_batch_and_skip_padding():
    model, _ = _conformer(True)
    lens = [400, 133, 36]
    x, la = synth_features(3, 400, lens=lens, seed=5)
    model.set_ffn_split(4)
    p0 = model.get_encoder_out(x, la)
    model.set_skip_padding(True)
    p1 = model.get_encoder_out(x, la)
    model.set_skip_padding(False)
    model.set_ffn_split(-1)
    for b, ln in enumerate(lens):
        nv = min(p0.shape[1], (ln + 3) // 4)
        assert torch.equal(p0[b, :nv], p1[b, :nv]) and not bool(p1[b, nv:].any())


def test_forced_split_on_a_batch_above_the_fused_attention_threshold():
    """> 128 row blocks: the default takes the fused kernels (attention reading the values in fragment order, which only
    the fused QKV stage writes).  A FORCED split must not pair the split QKV kernel with that attention kernel
    (tools/fuzz_split.py found logits off by 0.27 when it did): every mode within 1e-5 of the fused route."""
    model, _ = _conformer(True)
    B, T = 20, 1159
    lens = [T] + [int(v) for v in np.random.Generator(np.random.PCG64(7)).integers(200, T, size=B - 1)]
    x, la = synth_features(B, T, lens=lens, seed=11)
    assert B * ((T - 3) // 4) > 128 * 32
    model.set_ffn_split(0)
    ref = model.get_encoder_out(x, la, return_logits=True)[1]
    for mode in (2, 8, -1):
        model.set_ffn_split(mode)
        got = model.get_encoder_out(x, la, return_logits=True)[1]
        torch.cuda.synchronize()
        assert _rel(got.cpu().numpy(), ref.cpu().numpy()) < 1e-5, mode
    model.set_ffn_split(-1)
