"""GPU edge cases of the Conformer path: minimum / ragged / long inputs, empty utterances, determinism and
size-independent properties at the BASELINE shape (B=32, T=1000, V=4233)."""
import numpy as np
import pytest
import torch

from oracle.conformer_oracle import ConformerOracle
from oracle.ctc_decoders_oracle import greedy_tokens
from ppasr_amd.utils.synth import conformer_state_dict, synth_features

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _model(sd, V, L):
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
    return ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")


@pytest.mark.parametrize("B,T,lens", [
    (1, 7, [7]),                 # one output frame (the conv front-end's receptive field)
    (2, 11, [11, 7]),            # two frames
    (3, 131, [131, 1, 0]),       # lengths 1 and 0: every key masked -> fully masked attention rows -> zeros
    (1, 3000, [3000]),           # 30 s: T' = 749 (6 key blocks)
    (5, 67, [67, 66, 65, 64, 63]),
])
def test_ragged_and_extreme_lengths_match_oracle(B, T, lens):
    L, V = 2, 97
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=101, perturb_norm=True)
    x, lens = synth_features(B, T, lens=lens, seed=102)
    model = _model(sd, V, L)
    probs, logits = model.get_encoder_out(x, lens, return_logits=True)
    torch.cuda.synchronize()
    ref_probs, ref_logits = ConformerOracle(sd, num_blocks=L).get_encoder_out(x, lens, return_logits=True)
    assert tuple(probs.shape) == tuple(ref_probs.shape)
    assert torch.isfinite(probs).all()
    assert _rel(logits.cpu().numpy(), ref_logits.numpy()) < TOL


def test_too_short_input_is_refused():
    from ppasr_amd import _lib
    sd = conformer_state_dict(vocab_size=50, num_blocks=1, seed=1)
    model = _model(sd, 50, 1)
    with pytest.raises(_lib.PPASRHipError):
        model.get_encoder_out(np.zeros((1, 6, 80), np.float32), [6])


def test_baseline_shape_properties():
    """B=32 x 1000 frames, V=4233, 12 blocks (the bench workload): (i) bit-identical across runs, (ii) utterance
    independence: every utterance decoded alone gives the same tokens as inside the batch (no cross-utterance
    state: LayerNorm / global CMVN only), (iii) the first utterances match the CPU oracle token for token."""
    L, V, B, T = 12, 4233, 32, 1000
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=1234)
    x, lens = synth_features(B, T, seed=20440)
    model = _model(sd, V, L)
    t1, n1, s1 = model.encode_greedy(x, lens)
    t2, n2, s2 = model.encode_greedy(x, lens)
    torch.cuda.synchronize()
    assert torch.equal(t1, t2) and torch.equal(n1, n2) and torch.equal(s1, s2)
    for b in (0, 7, 31):
        # a single utterance is an under-filled launch and takes the split route by default (ppasr_set_ffn_split): same
        # arithmetic up to the order of the sum over feed-forward hidden chunks -> same tokens, scores to round-off
        tb, nb, sb = model.encode_greedy(x[b:b + 1], lens[b:b + 1])
        assert torch.equal(tb[0, :int(nb[0])], t1[b, :int(n1[b])])
        assert abs(float(sb[0]) - float(s1[b])) < 1e-3
        # on the same (fused) route: the attention walks the keys from the utterance's first row rounded down to a
        # multiple of 4 rows of the BATCH (aligned V^T loads), so the split of the keys over 64-key sub-blocks -- the
        # order of the softmax sums -- depends on (b * frames) % 4: bit-identical where that is 0, round-off elsewhere
        model.set_ffn_split(0)
        tb, nb, sb = model.encode_greedy(x[b:b + 1], lens[b:b + 1])
        model.set_ffn_split(-1)
        assert torch.equal(tb[0, :int(nb[0])], t1[b, :int(n1[b])])
        assert abs(float(sb[0]) - float(s1[b])) < (1e-9 if (b * 249) % 4 == 0 else 1e-4)
    ref = ConformerOracle(sd, num_blocks=L).get_encoder_out(x[:2], lens[:2])
    for b in range(2):
        ids, _, max_prob = greedy_tokens(ref[b].numpy())
        assert np.array_equal(ids, t1[b, :int(n1[b])].cpu().numpy())
        assert abs(float(s1[b]) - float(np.mean(max_prob.astype(np.float64))) * 100) < 1e-3


def test_padding_invariance_of_valid_frames():
    """Garbage in feature rows that only feed masked frames leaves the valid frames bit-identical
    (same property as tests/test_oracle.py::test_masked_frames_do_not_influence_valid_frames)."""
    L, V = 2, 64
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=77, perturb_norm=True)
    x, lens = synth_features(2, 99, lens=[99, 61], seed=78)
    model = _model(sd, V, L)
    n_valid = (61 + 3) // 4
    base = model.get_encoder_out(x, lens)[1].cpu().numpy()
    x2 = x.copy()
    x2[1, 4 * (n_valid - 1) + 7:] = 123.0
    pert = model.get_encoder_out(x2, lens)[1].cpu().numpy()
    assert np.array_equal(base[:n_valid], pert[:n_valid])
    assert not np.array_equal(base[n_valid:], pert[n_valid:])


def test_maximum_length_and_one_past_it():
    """The positional table holds max_len = 5000 positions (embedding.py:27-53,64-66): T' = 4999 is the longest
    utterance the reference accepts (200 s); one more output frame is refused."""
    from ppasr_amd import _lib
    L, V = 1, 61
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=131, perturb_norm=True)
    model = _model(sd, V, L)
    x, lens = synth_features(1, 19999, seed=132)
    assert model.out_frames(19999) == 4999
    probs, logits = model.get_encoder_out(x, lens, return_logits=True)
    torch.cuda.synchronize()
    ref_probs, ref_logits = ConformerOracle(sd, num_blocks=L).get_encoder_out(x, lens, return_logits=True)
    assert _rel(logits.cpu().numpy(), ref_logits.numpy()) < TOL
    tokens, n_tokens, _ = model.encode_greedy(x, lens)
    ids, _, _ = greedy_tokens(ref_probs[0].numpy())
    assert np.array_equal(ids, tokens[0, : int(n_tokens[0])].cpu().numpy())
    x2, lens2 = synth_features(1, 20003, seed=133)
    assert model.out_frames(20003) == 5000
    with pytest.raises(_lib.PPASRHipError):
        model.get_encoder_out(x2, lens2)


@pytest.mark.parametrize("input_layer,t_min,rate", [("conv2d6", 11, 6), ("conv2d8", 15, 8)])
def test_wider_front_ends_edges(input_layer, t_min, rate):
    """input_layer conv2d6 / conv2d8 (Conv2dSubsampling6 / 8, subsampling.py:118-205): the shortest input gives one frame,
    one frame less is refused, ragged lengths follow the oracle (masks by rate * t < len), a chunk goes through
    forward_chunk (chunk parity with the reference's source: tests/test_ref_pin_gpu.py, conf6 / conf8)."""
    from oracle.conformer_oracle import ConformerOracle
    from ppasr_amd import _lib
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    V, L = 50, 1
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=3, input_layer=input_layer)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                input_layer=input_layer)
    model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    assert model.subsampling_rate == rate and model.out_frames(t_min) == 1 and model.out_frames(t_min - 1) == 0
    with pytest.raises(_lib.PPASRHipError):
        model.get_encoder_out(np.zeros((1, t_min - 1, 80), np.float32), [t_min - 1])
    oracle = ConformerOracle(sd, num_blocks=L)
    for B, T, lens in ((1, t_min, [t_min]), (3, 157, [157, 80, 9]), (2, 1000, [1000, 333])):
        x, la = synth_features(B, T, lens=lens, seed=T)
        probs, logits = model.get_encoder_out(x, la, return_logits=True)
        ref_probs, ref_logits = oracle.get_encoder_out(x, la, return_logits=True)
        torch.cuda.synchronize()
        assert tuple(probs.shape) == tuple(ref_probs.shape)
        assert _rel(logits.cpu().numpy(), ref_logits.numpy()) < TOL
        nv = model.valid_out_frames(la, T).cpu().numpy()
        assert list(nv) == [min(probs.shape[1], (ln + rate - 1) // rate) for ln in lens]
    x, _ = synth_features(1, 67, seed=3)
    p, att, cnn = model.get_encoder_out_chunk(x, 0, -1)
    ref_p, _, _ = oracle.get_encoder_out_chunk(x, 0, -1)
    assert tuple(p.shape) == tuple(ref_p.shape) == (1, model.out_frames(67), V)
    assert _rel(p.cpu().numpy(), ref_p.numpy()) < TOL


@pytest.mark.parametrize("streaming,norm", [(True, "layer_norm"), (False, "batch_norm")])
def test_generic_width_512_with_8_heads(streaming, norm):
    """output_size 512 / attention_heads 8 (capi_generic.hip: dense layers + k_attention with 8 heads + row kernels):
    logits vs the oracle on ragged batches (lengths 1 and 0 included, > 128 keys), the fused greedy call, chunks."""
    from oracle.conformer_oracle import ConformerOracle
    from ppasr_amd import _lib
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    V, L = 77, 2
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=5, output_size=512, attention_heads=8, perturb_norm=True,
                              cnn_module_norm=norm)
    conf = dict(output_size=512, attention_heads=8, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                cnn_module_norm=norm)
    model = ConformerModel(80, V, streaming=streaming, encoder_conf=conf, state_dict=sd, device="cuda:0")
    oracle = ConformerOracle(sd, num_blocks=L, causal=streaming, attention_heads=8)
    for B, T, lens in ((1, 7, [7]), (3, 131, [131, 1, 0]), (2, 700, [700, 333])):
        x, la = synth_features(B, T, lens=lens, seed=T)
        probs, logits = model.get_encoder_out(x, la, return_logits=True)
        ref_probs, ref_logits = oracle.get_encoder_out(x, la, return_logits=True)
        tokens, n, _ = model.encode_greedy(x, la)
        torch.cuda.synchronize()
        assert torch.isfinite(probs).all()
        assert _rel(logits.cpu().numpy(), ref_logits.numpy()) < TOL
        for b in range(B):
            ids, _, _ = greedy_tokens(ref_probs[b].numpy())
            assert np.array_equal(ids, tokens[b, :int(n[b])].cpu().numpy()), b
    if streaming:  # forward_chunk on the general route: the stateless signature, caches through the reference layouts
        x, _ = synth_features(1, 67 + 64 + 40, seed=11)
        att = cnn = r_att = r_cnn = None
        off = 0
        for (a, b) in ((0, 67), (64, 131), (128, 171)):
            p, att, cnn = model.get_encoder_out_chunk(x[:, a:b], off, 20, att, cnn)
            rp, r_att, r_cnn = oracle.get_encoder_out_chunk(x[:, a:b], off, 20, r_att, r_cnn)
            off += p.shape[1]
            assert _rel(p.cpu().numpy(), rp.numpy()) < TOL
        assert tuple(att.shape) == tuple(r_att.shape) == (L, 8, 20, 128)
        assert _rel(att.cpu().numpy(), r_att.numpy()) < TOL and _rel(cnn.cpu().numpy(), r_cnn.numpy()) < TOL
        with pytest.raises(_lib.PPASRHipError):  # session groups stay on the fused route
            from ppasr_amd.model_utils.conformer.model import ConformerStreamGroup
            ConformerStreamGroup(model, 2)
