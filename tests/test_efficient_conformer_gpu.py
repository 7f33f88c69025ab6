"""GPU parity: HIP Efficient-Conformer path (grouped attention, stride-2 conv layer) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle.ctc_decoders_oracle import greedy_tokens
from oracle.efficient_conformer_oracle import EfficientConformerOracle
from ppasr_amd.utils.synth import efficient_conformer_state_dict, synth_features

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _model(sd, V, L, stride_idx, groups):
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                cnn_module_norm="layer_norm",
                efficient_conf=dict(stride_layer_idx=[stride_idx] if stride_idx is not None else [],
                                    stride=[2] if stride_idx is not None else [], group_layer_idx=list(groups),
                                    group_size=3, stride_kernel=True))
    return EfficientConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")


@pytest.mark.parametrize("B,T,lens,L,stride_idx,groups", [
    (3, 203, [203, 150, 67], 4, 1, (0, 1)),     # T'=49 (not a multiple of 3): zero-padded groups; odd -> ceil on stride
    (2, 1000, [1000, 700], 3, 2, (0, 1, 2)),    # T'=249 = 3*83
    (2, 411, [411, 300], 3, None, (1,)),        # no stride layer
    (2, 207, [207, 101], 3, 0, ()),             # stride only
])
def test_efficient_conformer_matches_oracle(B, T, lens, L, stride_idx, groups):
    V = 300
    sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=71, perturb_norm=True,
                                        stride_layer_idx=stride_idx, group_layer_idx=groups)
    x, lens = synth_features(B, T, lens=lens, seed=72)
    model = _model(sd, V, L, stride_idx, groups)
    probs, logits = model.get_encoder_out(x, lens, return_logits=True)
    torch.cuda.synchronize()
    oracle = EfficientConformerOracle(sd, num_blocks=L, stride_layer_idx=stride_idx, group_layer_idx=groups)
    ref_probs, ref_logits = oracle.get_encoder_out(x, lens, return_logits=True)
    assert tuple(logits.shape) == tuple(ref_logits.shape)
    e = _rel(logits.cpu().numpy(), ref_logits.numpy())
    print("logits", e)
    assert e < TOL
    tokens, n_tokens, score = model.encode_greedy(x, lens)
    for b in range(B):
        ids, _, _ = greedy_tokens(ref_probs[b].numpy())
        assert np.array_equal(ids, tokens[b, : int(n_tokens[b])].cpu().numpy())


def test_efficient_conformer_full_config_beam_search():
    """configs[3]: Efficient-Conformer (12 blocks, stride at 3, groups 0-3), beam 10, cutoff 0.99 / top-40."""
    from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
    V, L = 4233, 12
    sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=81)
    x, lens = synth_features(2, 503, lens=[503, 400], seed=82)
    model = _model(sd, V, L, 3, (0, 1, 2, 3))
    probs, logits = model.get_encoder_out(x, lens, return_logits=True)
    ref_probs, ref_logits = EfficientConformerOracle(sd, num_blocks=L).get_encoder_out(x, lens, return_logits=True)
    assert probs.shape[1] == 63  # T'=125 -> ceil(125/2)
    assert _rel(logits.cpu().numpy(), ref_logits.numpy()) < TOL
    tokens, ln, sc, _ = beam_search_ids(probs, 10, 0.99, 40, 0)
    assert int(ln[0, 0]) > 0


@pytest.mark.parametrize("B,T,lens", [(3, 203, [203, 150, 67]), (2, 411, [411, 300]), (1, 131, [131])])
@pytest.mark.parametrize("route", [-1, 0])
def test_efficient_conformer_non_streaming_matches_oracle(B, T, lens, route):
    """streaming=False: non-causal conv modules (depthwise padding (k-1)//2 on both sides, the strided one included,
    efficient_conformer/convolution.py ctor), full attention as before."""
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    V, L = 113, 4
    sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=B * 100 + T, perturb_norm=True, stride_layer_idx=1,
                                        group_layer_idx=(0, 1))
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                cnn_module_norm="layer_norm",
                efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3,
                                    stride_kernel=True))
    model = EfficientConformerModel(80, V, streaming=False, encoder_conf=conf, state_dict=sd, device="cuda:0")
    model.set_ffn_split(route)
    oracle = EfficientConformerOracle(sd, num_blocks=L, stride_layer_idx=1, group_layer_idx=(0, 1), causal=False)
    x, la = synth_features(B, T, lens=lens, seed=T)
    probs, logits = model.get_encoder_out(x, la, return_logits=True)
    ref_probs, ref_logits = oracle.get_encoder_out(x, la, return_logits=True)
    torch.cuda.synchronize()
    assert tuple(logits.shape) == tuple(ref_logits.shape)
    assert _rel(logits.cpu().numpy(), ref_logits.numpy()) < TOL
    # the non-causal modules must actually differ from the causal ones
    causal = EfficientConformerOracle(sd, num_blocks=L, stride_layer_idx=1, group_layer_idx=(0, 1)).get_encoder_out(
        x, la, return_logits=True)[1]
    assert _rel(causal.numpy(), ref_logits.numpy()) > 1e-2
    with pytest.raises(Exception):
        model.new_stream()  # forward_chunk needs the causal module
