"""GPU parity of the streaming path (forward_chunk with device-resident caches) vs the oracle's
restatement of conformer/encoder.py:208-283, chunk by chunk, with predict_stream's windowing
(67-frame windows, stride 64, required_cache_size = -16: predict.py:277-283,306-307)."""
import numpy as np
import pytest
import torch

from oracle.conformer_oracle import ConformerOracle
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


def _windows(n_frames, window=67, stride=64):
    out = []
    for cur in range(0, n_frames - 7 + 1, stride):
        out.append((cur, min(cur + window, n_frames)))
    return out


@pytest.mark.parametrize("required", [-16, 32])
def test_chunked_stream_matches_oracle(required):
    L, V = 2, 200
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=31, perturb_norm=True)
    x, _ = synth_features(1, 64 * 5 + 30, seed=32)
    model = _model(sd, V, L)
    oracle = ConformerOracle(sd, num_blocks=L)
    stream = model.new_stream()
    att = cnn = None
    offset = 0
    for (a, b) in _windows(x.shape[1]):
        chunk = x[:, a:b]
        ref, att, cnn = oracle.get_encoder_out_chunk(chunk, offset, required, att, cnn)
        got = stream.encode_chunk(chunk, required)
        torch.cuda.synchronize()
        assert got.shape == ref.shape
        assert _rel(got.cpu().numpy(), ref.numpy()) < TOL, (a, b)
        offset += ref.shape[1]
        assert stream.offset == offset and stream.cache_frames == att.shape[2]
        g_att, g_cnn = stream.export_caches()
        assert _rel(g_att.cpu().numpy(), att.numpy()) < TOL
        assert _rel(g_cnn.cpu().numpy(), cnn.numpy()) < TOL
    # reset -> the first chunk again gives the first result again
    first_ref, _, _ = oracle.get_encoder_out_chunk(x[:, :67], 0, required)
    stream.reset()
    assert stream.offset == 0 and stream.cache_frames == 0
    again = stream.encode_chunk(x[:, :67], required)
    assert _rel(again.cpu().numpy(), first_ref.numpy()) < TOL


def test_stateless_chunk_signature_and_one_giant_chunk():
    """get_encoder_out_chunk(speech, offset, required, att_cache, cnn_cache) with explicit caches
    (conformer/model.py:164-184), and the reference's predict(): the whole utterance as ONE chunk
    with empty caches equals get_encoder_out (inference_predictor.py:127-137)."""
    L, V = 2, 150
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=41, perturb_norm=True)
    x, lens = synth_features(1, 407, seed=42)
    model = _model(sd, V, L)
    oracle = ConformerOracle(sd, num_blocks=L)
    full = model.get_encoder_out(x, lens)
    one, att, cnn = model.get_encoder_out_chunk(x, 0, -1)
    torch.cuda.synchronize()
    assert _rel(one.cpu().numpy(), full.cpu().numpy()) < 1e-5
    r_probs, r_att, r_cnn = oracle.get_encoder_out_chunk(x, 0, -1)
    assert _rel(att.cpu().numpy(), r_att.numpy()) < TOL and _rel(cnn.cpu().numpy(), r_cnn.numpy()) < TOL
    # continue from explicitly passed (host) caches
    x2, _ = synth_features(1, 67, seed=43)
    ref2, _, _ = oracle.get_encoder_out_chunk(x2, r_probs.shape[1], -1, r_att, r_cnn)
    got2, att2, _ = model.get_encoder_out_chunk(x2, r_probs.shape[1], -1, r_att, r_cnn)
    torch.cuda.synchronize()
    assert _rel(got2.cpu().numpy(), ref2.numpy()) < TOL
    assert att2.shape[2] == r_att.shape[2] + 16


def test_session_group_equals_independent_streams():
    """ppasr_encode_chunk_group: sessions that start at different times (different offsets / cache lengths) advanced with
    one set of launches give exactly what independent single-session streams give."""
    from ppasr_amd.model_utils.conformer.model import ConformerStreamGroup
    L, V = 2, 150
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=61, perturb_norm=True)
    model = _model(sd, V, L)
    n_sessions, n_rounds = 5, 6
    feats = [synth_features(1, 64 * n_rounds + 3, seed=70 + s)[0] for s in range(n_sessions)]
    group = ConformerStreamGroup(model, n_sessions, max_frames=256)
    singles = [model.new_stream() for _ in range(n_sessions)]
    start = [0, 0, 1, 2, 4]  # round in which each session joins
    for r in range(n_rounds):
        active = [s for s in range(n_sessions) if r >= start[s]]
        if r == 3:
            active.remove(1)  # a session may skip a round
        chunks = np.concatenate([feats[s][:, 64 * (r - start[s]) - (64 if (s == 1 and r > 3) else 0):][:, :67]
                                 for s in active], axis=0)
        fa, fp, probs = group.encode_chunks(active, chunks, want_probs=True)
        torch.cuda.synchronize()
        for i, s in enumerate(active):
            ref = singles[s].encode_chunk(chunks[i:i + 1], -16)
            torch.cuda.synchronize()
            assert group.offset(s) == singles[s].offset
            err = _rel(probs[i].cpu().numpy(), ref[0].cpu().numpy())
            # (not bit-equal: the single-session route runs 16-row units and splits the front end's contractions over
            #  more workgroups than the group route -- another order of the fp32 sums; both sit within TOL of the oracle)
            assert err < 3e-5, (r, s, err)
            assert np.array_equal(fa[i].cpu().numpy(), ref[0].argmax(dim=1).cpu().numpy())
    # reset one session: it restarts from offset 0 while the others keep their state
    group.reset(2)
    singles[2].reset()
    fa, fp, probs = group.encode_chunks([2, 0], np.concatenate([feats[2][:, :67], feats[0][:, 64 * n_rounds - 64:][:, :67]]),
                                        want_probs=True)
    ref2 = singles[2].encode_chunk(feats[2][:, :67], -16)
    torch.cuda.synchronize()
    assert group.offset(2) == 16 and _rel(probs[0].cpu().numpy(), ref2[0].cpu().numpy()) < 3e-5
    with pytest.raises(Exception):
        group.encode_chunks([1, 1], np.concatenate([feats[1][:, :67]] * 2))
