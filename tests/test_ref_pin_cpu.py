"""CPU: the oracles (oracle/*_oracle.py, the builder's restatements) against fixtures produced by the REFERENCE's OWN
model source (tests/golden/ref_small.npz, written by tests/golden/make_ref_goldens.py which imports
/root/reference/ppasr/model_utils/*/model.py unmodified on top of oracle/paddle_shim).  This is what pins the oracles:
both sides are fp32 on torch CPU kernels, so they must agree to accumulation-order round-off (tolerance 2e-5 relative
to the tensor's largest magnitude; measured values are printed)."""
import os

import numpy as np
import pytest
import torch

import ref_cases as rc
from oracle.conformer_oracle import ConformerOracle
from oracle.deepspeech2_oracle import DeepSpeech2Oracle
from oracle.efficient_conformer_oracle import EfficientConformerOracle
from oracle.squeezeformer_oracle import SqueezeformerOracle

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 2e-5


@pytest.fixture(scope="module")
def ref():
    with np.load(os.path.join(HERE, "golden", "ref_small.npz")) as z:
        return {k: z[k] for k in z.files}


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def make_oracle(case, sd):
    fam, L, kw = case["family"], case["L"], case["kw"]
    causal = case["streaming"]
    if fam == "conformer":
        opts = {k: kw[k] for k in ("pos_enc_layer_type", "normalize_before", "concat_after", "macaron_style",
                                   "use_cnn_module", "activation_type", "cnn_module_kernel") if k in kw}
        return ConformerOracle(sd, num_blocks=L, causal=causal, attention_heads=kw.get("attention_heads", 4), **opts)
    if fam == "efficient_conformer":
        return EfficientConformerOracle(sd, num_blocks=L, stride_layer_idx=kw["stride_layer_idx"],
                                        group_layer_idx=kw["group_layer_idx"], group_size=kw.get("group_size", 3), causal=causal,
                                        attention_heads=kw.get("attention_heads", 4),
                                        cnn_module_kernel=kw.get("cnn_module_kernel", 15))
    if fam == "squeezeformer":
        return SqueezeformerOracle(sd, num_blocks=L, reduce_idx=kw["reduce_idx"], recover_idx=kw["recover_idx"], causal=causal,
                                   attention_heads=kw.get("attention_heads", 4), adaptive_scale=kw.get("adaptive_scale", True),
                                   activation_type=kw.get("activation_type", "swish"),
                                   normalize_before=kw.get("normalize_before", False),
                                   pos_enc_layer_type=kw.get("pos_enc_layer_type", "rel_pos"))
    return DeepSpeech2Oracle(sd, num_rnn_layers=L, streaming=case["streaming"], use_gru=kw.get("use_gru", False))


FORMERS = [k for k, c in rc.SMALL.items() if c["family"] != "deepspeech2"]
DS2 = [k for k, c in rc.SMALL.items() if c["family"] == "deepspeech2"]


@pytest.mark.parametrize("name", FORMERS)
def test_former_oracle_matches_reference_source(ref, name):
    case = rc.SMALL[name]
    sd = rc.state_dict(case)
    oracle = make_oracle(case, sd)
    x, lens = rc.features(case)
    probs, logits = oracle.get_encoder_out(x, lens, return_logits=True)
    e_l, e_p = _rel(logits.numpy(), ref[f"{name}/logits"]), _rel(probs.numpy(), ref[f"{name}/probs"])
    print(f"{name}: logits {e_l:.2e} probs {e_p:.2e}")
    assert e_l < TOL and e_p < TOL
    # greedy ids of every frame (padded ones too, like greedy_decoder_batch) are identical
    assert np.array_equal(probs.numpy().argmax(-1), ref[f"{name}/probs"].argmax(-1))


@pytest.mark.parametrize("name,required", [(k, r) for k in FORMERS for r in rc.SMALL[k]["required"]])
def test_former_oracle_chunks_match_reference_source(ref, name, required):
    case = rc.SMALL[name]
    sd = rc.state_dict(case)
    oracle = make_oracle(case, sd)
    x = rc.chunk_features(case)
    att = cnn = None
    offset, outs = 0, []
    for (a, b) in rc.windows(x.shape[1]):
        p, att, cnn = oracle.get_encoder_out_chunk(x[:, a:b], offset, required, att, cnn)
        outs.append(p.numpy())
        offset += p.shape[1]
    k = f"{name}/chunk{required}"
    assert [o.shape[1] for o in outs] == ref[k + "/n"].tolist()
    e_p = _rel(np.concatenate(outs, 1), ref[k + "/probs"])
    e_a = _rel(att.numpy(), ref[k + "/att"]) if ref[k + "/att"].size else 0.0
    e_c = _rel(cnn.numpy(), ref[k + "/cnn"]) if ref[k + "/cnn"].size else 0.0  # (use_cnn_module=False: empty)
    print(f"{k}: probs {e_p:.2e} att {e_a:.2e} cnn {e_c:.2e}")
    assert tuple(att.shape) == ref[k + "/att"].shape and tuple(cnn.shape) == ref[k + "/cnn"].shape
    assert e_p < TOL and e_a < TOL and e_c < TOL


@pytest.mark.parametrize("name", DS2)
def test_ds2_oracle_matches_reference_source(ref, name):
    case = rc.SMALL[name]
    sd = rc.state_dict(case)
    oracle = make_oracle(case, sd)
    x, lens = rc.features(case)
    probs, out_lens, h, c = oracle.forward(x, lens)
    e = _rel(probs.numpy(), ref[f"{name}/probs"])
    print(f"{name}: probs {e:.2e}")
    assert e < TOL
    if case["chunk_frames"]:
        xc = rc.chunk_features(case)
        B = xc.shape[0]
        h = torch.zeros(case["L"], B, 1024)
        c = torch.zeros(case["L"], B, 1024)
        outs = []
        for (a, b) in rc.windows(xc.shape[1]):
            p, _, h, c = oracle.forward(xc[:, a:b], np.full(B, b - a, np.int64), h, c)
            outs.append(p.numpy())
        assert [o.shape[1] for o in outs] == ref[f"{name}/chunk/n"].tolist()
        e_p, e_h = _rel(np.concatenate(outs, 1), ref[f"{name}/chunk/probs"]), _rel(h.numpy(), ref[f"{name}/chunk/h"])
        print(f"{name}/chunk: probs {e_p:.2e} h {e_h:.2e}")
        assert e_p < TOL and e_h < TOL
        if not case["kw"].get("use_gru"):
            assert _rel(c.numpy(), ref[f"{name}/chunk/c"]) < TOL


def test_full_size_conformer_oracle_matches_reference_source():
    """configs[1] (12 blocks, T = 1000, V = 4233): the oracle on utterances 0-1 vs the reference-source summary of
    ref_full.npz (utterances of equal length do not interact, so a sub-batch reproduces the batch's rows)."""
    with np.load(os.path.join(HERE, "golden", "ref_full.npz")) as z:
        ids, margin, lse, sampled, cols = (z["cfg2/" + k] for k in ("ids", "margin", "lse", "sampled", "cols"))
    case = rc.FULL["cfg2"]
    sd = rc.state_dict(case)
    x, lens = rc.features(case)
    _, logits = ConformerOracle(sd, num_blocks=12).get_encoder_out(x[:2], lens[:2], return_logits=True)
    lg = logits.numpy().astype(np.float64)
    e_s = float(np.abs(lg[..., cols] - sampled[:2]).max() / np.abs(sampled[:2]).max())
    m = lg.max(-1, keepdims=True)
    e_z = float(np.abs((m[..., 0] + np.log(np.exp(lg - m).sum(-1))) - lse[:2]).max() / np.abs(lse[:2]).max())
    print(f"cfg2 oracle vs reference source: sampled logits {e_s:.2e} lse {e_z:.2e}")
    assert e_s < TOL and e_z < TOL
    clear = margin[:2] > 1e-3
    assert np.array_equal(lg.argmax(-1)[clear], ids[:2][clear])


def test_full_size_deepspeech2_oracle_matches_reference_source():
    """configs[0] (5 x 1024 bidirectional LSTM, B = 1, T = 498, V = 4233): oracle/deepspeech2_oracle.py vs the
    reference-source summary of ref_full.npz (probabilities on the sampled columns = exp(logit - lse), greedy ids)."""
    with np.load(os.path.join(HERE, "golden", "ref_full.npz")) as z:
        ids, margin, lse, sampled, cols, maxprob = (z["cfg1/" + k] for k in ("ids", "margin", "lse", "sampled", "cols", "maxprob"))
    case = rc.FULL["cfg1"]
    sd = rc.state_dict(case)
    x, lens = rc.features(case)
    probs, out_lens, _, _ = make_oracle(case, sd).forward(x, lens)
    p = probs.numpy().astype(np.float64)
    assert p.shape == (1, 123, 4233) and out_lens.tolist() == [123]
    want = np.exp(sampled.astype(np.float64) - lse.astype(np.float64)[..., None])
    e_p = float(np.abs(p[..., cols] - want).max() / want.max())
    e_m = float(np.abs(p.max(-1) - maxprob).max())
    print(f"cfg1 oracle vs reference source: sampled probs {e_p:.2e} max-prob {e_m:.2e}")
    assert e_p < TOL and e_m < TOL
    clear = margin > 1e-3
    assert np.array_equal(p.argmax(-1)[clear], ids[clear])


def test_reference_wav_oracle_chain_matches_reference_predictor_source():
    """tests/golden/ref_wav.npz (the reference's PPASRPredictor source on /root/reference/dataset/test.wav): the oracle chain
    fbank_oracle.featurize -> ConformerOracle -> ctc_decoders_oracle.greedy_tokens reproduces the reference's features
    (the reference's AudioSegment.normalize / to('int16') source ran in front of the same fbank arithmetic), frame ids and
    text."""
    from oracle import fbank_oracle
    from oracle.ctc_decoders_oracle import greedy_tokens
    from ppasr_amd.utils.synth import conformer_state_dict, synth_vocabulary
    with np.load(os.path.join(HERE, "golden", "ref_wav.npz")) as z:
        z = {k: z[k] for k in z.files}
    V = int(z["vocab_size"])
    feats = fbank_oracle.featurize(z["samples"].astype(np.float32) / 32768.0, int(z["sample_rate"])).astype(np.float32)
    assert feats.shape[0] == int(z["n_feature_frames"])
    assert float(np.abs(feats[::16] - z["feats_16"]).max()) < 1e-4
    sd = conformer_state_dict(vocab_size=V, num_blocks=12, seed=int(z["sd_seed"]))
    probs = ConformerOracle(sd, num_blocks=12).get_encoder_out(feats[None], np.array([feats.shape[0]]))[0].numpy()
    ids = probs.argmax(-1)
    clear = z["margin"] > 1e-4
    assert np.array_equal(ids[clear], z["ids"][clear])
    assert float(np.abs(probs.max(-1) - z["maxprob"]).max()) < TOL
    tok, _, _ = greedy_tokens(probs)
    vocab = synth_vocabulary(V)
    if clear.all():
        assert "".join(vocab[i] for i in tok).replace("<space>", " ") == str(z["predict_text"])


def test_reference_wav_stream_windows_with_in_place_normalisation():
    """predict_stream's state machine (predict.py:232-337) restated with the fbank oracle: per call the buffered samples are
    featurised, NORMALISED IN PLACE (the reference's AudioSegment.normalize mutates `remained_wav`), cut by 160 x frames;
    windows of 67 frames, stride 64, 3 cached frames.  The windows must be the ones the reference's own source fed to its
    model (sizes, first rows, float64 sums from tests/golden/ref_wav.npz)."""
    from oracle import fbank_oracle
    from ppasr_amd.data_utils.featurizer import db_gain
    with np.load(os.path.join(HERE, "golden", "ref_wav.npz")) as z:
        z = {k: z[k] for k in z.files}
    x = z["samples"].astype(np.float32) / 32768.0
    step = int(int(z["sample_rate"]) * float(z["chunk_seconds"]))
    remained, cached, wins = None, None, []
    for i in range(0, len(x), step):
        is_end = i + step >= len(x)
        remained = x[i:i + step] if remained is None else np.concatenate([remained, x[i:i + step]])
        feat = fbank_oracle.featurize(remained, int(z["sample_rate"])).astype(np.float32)
        remained = remained * db_gain(remained, -20)
        cached = feat if cached is None else np.concatenate([cached, feat], 0)
        remained = remained[160 * feat.shape[0]:]
        n = cached.shape[0]
        if (n < 67 and not is_end) or n < 7:
            continue
        end = None
        for cur in range(0, n - (7 if is_end else 67) + 1, 64):
            end = min(cur + 67, n)
            wins.append(cached[cur:end])
        cached = cached[end - 3:]
    assert [w.shape[0] for w in wins] == z["stream_win_frames"].tolist()
    first = np.stack([w[0] for w in wins])
    assert float(np.abs(first - z["stream_win_first"]).max()) < 1e-4
    sums = np.array([w.astype(np.float64).sum() for w in wins])
    assert float(np.abs(sums - z["stream_win_sums"]).max()) < 0.05  # (5 360 values of magnitude ~10 per window)
