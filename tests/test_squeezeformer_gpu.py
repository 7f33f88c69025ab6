"""GPU parity: HIP Squeezeformer path vs the CPU oracle (squeezeformer_oracle.py), stage taps + logits."""
import numpy as np
import pytest
import torch

from oracle.ctc_decoders_oracle import greedy_tokens
from oracle.squeezeformer_oracle import SqueezeformerOracle
from ppasr_amd.utils.synth import squeezeformer_state_dict, synth_features

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _model(sd, V, L, reduce_idx, recover_idx):
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=reduce_idx,
                recover_idx=recover_idx, feed_forward_expansion_factor=8, cnn_module_kernel=31)
    return SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")


@pytest.mark.parametrize("B,T,lens,L,red,rec", [
    (3, 203, [203, 150, 67], 4, 1, 3),    # reduce before layer 1, recover before layer 3
    (2, 411, [411, 300], 3, None, None),  # no time reduction
    (2, 207, [207, 101], 5, 2, 4),        # odd T' (51): ceil(T'/2) reduced frames
])
def test_squeezeformer_layers_and_logits_match_oracle(B, T, lens, L, red, rec):
    V = 300
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=51, perturb_norm=True)
    x, lens = synth_features(B, T, lens=lens, seed=52)
    model = _model(sd, V, L, red, rec)
    probs, logits = model.get_encoder_out(x, lens, return_logits=True)
    torch.cuda.synchronize()
    oracle = SqueezeformerOracle(sd, num_blocks=L, reduce_idx=red, recover_idx=rec)
    with torch.no_grad():
        enc, _, layers = oracle.encoder_forward(x, lens, return_layers=True)
        ref_logits = oracle.ctc_logits(enc)
        ref_probs = torch.softmax(ref_logits, dim=2)
    e_logits = _rel(logits.cpu().numpy(), ref_logits.numpy())
    e_probs = _rel(probs.cpu().numpy(), ref_probs.numpy())
    print("logits", e_logits, "probs", e_probs)
    assert e_logits < TOL and e_probs < TOL
    tokens, n_tokens, score = model.encode_greedy(x, lens)
    for b in range(B):
        ids, _, _ = greedy_tokens(ref_probs[b].numpy())
        assert np.array_equal(ids, tokens[b, : int(n_tokens[b])].cpu().numpy())


def test_squeezeformer_full_config_runs():
    """configs/squeezeformer.yml shape: 12 blocks, reduce 5, recover 11, kernel 31, V=4233."""
    V, L = 4233, 12
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=61)
    x, lens = synth_features(2, 331, lens=[331, 200], seed=62)
    model = _model(sd, V, L, 5, 11)
    probs, logits = model.get_encoder_out(x, lens, return_logits=True)
    ref_probs, ref_logits = SqueezeformerOracle(sd, num_blocks=L).get_encoder_out(x, lens, return_logits=True)
    assert _rel(logits.cpu().numpy(), ref_logits.numpy()) < TOL


@pytest.mark.parametrize("B,T,lens", [(3, 203, [203, 150, 67]), (2, 411, [411, 300]), (1, 131, [131])])
@pytest.mark.parametrize("route", [-1, 0])
def test_squeezeformer_non_streaming_matches_oracle(B, T, lens, route):
    """streaming=False (squeezeformer/model.py:35-39): non-causal conv modules (depthwise padding 15 on both sides) and
    TimeReductionLayer1D (5-tap depthwise conv, stride 2, padding 3) instead of the 1-tap stream layer."""
    import numpy as np
    from oracle.squeezeformer_oracle import SqueezeformerOracle
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    from ppasr_amd.utils.synth import squeezeformer_state_dict, synth_features
    V, L = 131, 4
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=B * 100 + T, perturb_norm=True, streaming=False)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=1, recover_idx=3,
                feed_forward_expansion_factor=8, cnn_module_kernel=31)
    model = SqueezeformerModel(80, V, streaming=False, encoder_conf=conf, state_dict=sd, device="cuda:0")
    model.set_ffn_split(route)
    oracle = SqueezeformerOracle(sd, num_blocks=L, reduce_idx=1, recover_idx=3, causal=False)
    x, la = synth_features(B, T, lens=lens, seed=T)
    _, logits = model.get_encoder_out(x, la, return_logits=True)
    _, ref = oracle.get_encoder_out(x, la, return_logits=True)
    torch.cuda.synchronize()
    a, b = logits.cpu().numpy().astype(np.float64), ref.numpy().astype(np.float64)
    assert a.shape == b.shape and np.abs(a - b).max() / np.abs(b).max() < 1e-3
    with pytest.raises(Exception):
        model.new_stream()  # forward_chunk needs the causal module
