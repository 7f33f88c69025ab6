"""Seeded random sweep over batch shapes / lengths for every encoder family: logits within 1e-3 (relative to the largest
logit) of the oracle and greedy token ids bit-exact.  Complements the hand-picked cases of the per-family test files."""
import numpy as np
import pytest
import torch

from oracle.ctc_decoders_oracle import greedy_tokens
from ppasr_amd.utils.synth import (conformer_state_dict, deepspeech2_state_dict, efficient_conformer_state_dict,
                                   squeezeformer_state_dict, synth_features)

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _cases(seed, n, t_lo=40, t_hi=900):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for _ in range(n):
        B = int(rng.integers(1, 6))
        T = int(rng.integers(t_lo, t_hi))
        lens = sorted((int(x) for x in rng.integers(max(7, T // 5), T + 1, size=B)), reverse=True)
        lens[0] = T  # the batch is padded to its longest utterance (collate_fn.py:17)
        out.append((B, T, lens))
    return out


def _check(model, oracle, x, lens, V):
    probs, logits = model.get_encoder_out(x, lens, return_logits=True)
    torch.cuda.synchronize()
    ref_probs, ref_logits = oracle.get_encoder_out(x, lens, return_logits=True)
    assert tuple(logits.shape) == tuple(ref_logits.shape)
    assert _rel(logits.cpu().numpy(), ref_logits.numpy()) < TOL
    tokens, n_tokens, _ = model.encode_greedy(x, lens)
    for b in range(x.shape[0]):
        ids, _, _ = greedy_tokens(ref_probs[b].numpy())
        assert np.array_equal(ids, tokens[b, : int(n_tokens[b])].cpu().numpy())


@pytest.mark.parametrize("B,T,lens", _cases(101, 5))
def test_conformer_sweep(B, T, lens):
    from oracle.conformer_oracle import ConformerOracle
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    V, L = 97, 2
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=B * 1000 + T, perturb_norm=True)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
    x, lens = synth_features(B, T, lens=lens, seed=T)
    _check(ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0"),
           ConformerOracle(sd, num_blocks=L), x, lens, V)


@pytest.mark.parametrize("B,T,lens", _cases(202, 4))
def test_squeezeformer_sweep(B, T, lens):
    from oracle.squeezeformer_oracle import SqueezeformerOracle
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    V, L = 131, 4
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=B * 1000 + T, perturb_norm=True)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=1, recover_idx=3,
                feed_forward_expansion_factor=8, cnn_module_kernel=31)
    x, lens = synth_features(B, T, lens=lens, seed=T)
    _check(SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0"),
           SqueezeformerOracle(sd, num_blocks=L, reduce_idx=1, recover_idx=3), x, lens, V)


@pytest.mark.parametrize("B,T,lens", _cases(303, 4))
def test_efficient_conformer_sweep(B, T, lens):
    from oracle.efficient_conformer_oracle import EfficientConformerOracle
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    V, L = 113, 4
    sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=B * 1000 + T, perturb_norm=True,
                                        stride_layer_idx=1, group_layer_idx=(0, 1))
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                cnn_module_norm="layer_norm",
                efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3,
                                    stride_kernel=True))
    x, lens = synth_features(B, T, lens=lens, seed=T)
    _check(EfficientConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0"),
           EfficientConformerOracle(sd, num_blocks=L, stride_layer_idx=1, group_layer_idx=(0, 1)), x, lens, V)


@pytest.mark.parametrize("B,T,lens", _cases(404, 3, 40, 300))
@pytest.mark.parametrize("streaming", [True, False])
def test_deepspeech2_sweep(B, T, lens, streaming):
    from oracle.deepspeech2_oracle import DeepSpeech2Oracle
    from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
    V = 89
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=2, streaming=streaming, seed=B * 1000 + T)
    x, lens = synth_features(B, T, lens=lens, seed=T)
    model = DeepSpeech2Model(80, V, streaming=streaming, encoder_conf=dict(num_rnn_layers=2, rnn_size=1024), state_dict=sd)
    probs = model.get_encoder_out(x, lens)
    torch.cuda.synchronize()
    ref, _, _, _ = DeepSpeech2Oracle(sd, 2, 1024, streaming).forward(x, lens)
    assert _rel(probs.cpu().numpy(), np.asarray(ref)) < TOL
