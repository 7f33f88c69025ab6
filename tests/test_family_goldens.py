"""Committed fixtures of the non-Conformer paths (tests/golden/family_oracle_golden.npz, made by
tests/golden/make_family_goldens.py from the CPU restatements -- the reference itself cannot produce them offline).
CPU half: the oracles still reproduce them (drift guard).  GPU half: the HIP path against the fixtures, with no oracle
in the loop at run time."""
import importlib.util
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "family_oracle_golden.npz")
_spec = importlib.util.spec_from_file_location("make_family_goldens", os.path.join(HERE, "golden", "make_family_goldens.py"))
mk = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mk)


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_oracles_reproduce_the_committed_fixtures():
    g = np.load(GOLD)
    got = mk.compute(mk.cases())
    assert set(got) == set(g.files)
    for k in g.files:
        if k.endswith("_tokens") or k.endswith("_lens"):
            assert np.array_equal(got[k], g[k]), k
        elif k.endswith("_score"):
            assert abs(float(got[k]) - float(g[k])) <= 1e-6 * max(1.0, abs(float(g[k]))), k
        else:
            assert _rel(got[k], g[k]) < 1e-4, k


@pytest.mark.gpu
def test_squeezeformer_and_efficient_conformer_match_fixtures():
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    g, c = np.load(GOLD), mk.cases()
    m = SqueezeformerModel(80, 61, streaming=True, encoder_conf=c["sq_conf"], state_dict=c["sq_sd"], device="cuda:0")
    for mode in (-1, 0):
        m.set_ffn_split(mode)
        _, lg = m.get_encoder_out(c["sq_x"], c["sq_lens"], return_logits=True)
        assert _rel(lg.cpu().numpy(), g["sq_logits"]) < 1e-3, mode
    m = EfficientConformerModel(80, 53, streaming=True, encoder_conf=c["eff_conf"], state_dict=c["eff_sd"], device="cuda:0")
    for mode in (-1, 0):
        m.set_ffn_split(mode)
        _, lg = m.get_encoder_out(c["eff_x"], c["eff_lens"], return_logits=True)
        assert _rel(lg.cpu().numpy(), g["eff_logits"]) < 1e-3, mode


@pytest.mark.gpu
@pytest.mark.parametrize("key,streaming", [("ds2s", True), ("ds2b", False)])
def test_deepspeech2_matches_fixtures(key, streaming):
    from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
    g, c = np.load(GOLD), mk.cases()
    m = DeepSpeech2Model(80, 47, streaming=streaming, encoder_conf=dict(num_rnn_layers=2, rnn_size=1024),
                         state_dict=c[key + "_sd"], device="cuda:0")
    probs, lens, h, _ = m.get_encoder_out_chunk(c[key + "_x"], c[key + "_lens"])
    assert lens.cpu().tolist() == g[key + "_lens"].tolist()
    assert _rel(probs.cpu().numpy(), g[key + "_probs"]) < 1e-3
    assert _rel(h.cpu().numpy(), g[key + "_h"]) < 1e-3


@pytest.mark.gpu
def test_beam_search_and_fbank_match_fixtures():
    from ppasr_amd.data_utils.featurizer import AudioFeaturizer
    from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
    g, c = np.load(GOLD), mk.cases()
    for i, (p, beam, cp, tn) in enumerate(c["beam"]):
        tokens, lens, scores, _ = beam_search_ids(torch.from_numpy(p)[None].cuda(), beam, cp, tn, 0, nbest=1)
        n = int(lens[0, 0])
        assert np.array_equal(tokens[0, 0, :n].cpu().numpy(), g[f"beam{i}_tokens"]), i
        assert abs(float(scores[0, 0]) - float(g[f"beam{i}_score"])) < 1e-3 * max(1.0, abs(float(g[f"beam{i}_score"])))
    f = AudioFeaturizer(feature_method="fbank", n_mels=80, sample_rate=16000, use_dB_normalization=True, target_dB=-20)
    feat = f.featurize(c["wav"])
    assert feat.shape == g["fbank"].shape and float(np.abs(feat - g["fbank"]).max()) < 1e-3
