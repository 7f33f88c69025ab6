"""CPU tests of the oracles: pinned against committed golden vectors.

* greedy decoder restatement == the REFERENCE's own ctc_greedy_decoder.py outputs
  (tests/golden/ctc_greedy_golden.npz, made by importing the reference, see make_goldens.py)
* Conformer oracle == its own committed outputs (drift guard; encoder parity is unpinned
  by the reference) + structural self-checks taken from the reference's semantics.
"""
import os

import numpy as np
import torch

from oracle import ctc_decoders_oracle as dec
from oracle.conformer_oracle import ConformerOracle
from ppasr_amd.utils.synth import conformer_state_dict, synth_features

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _vocab(V):
    return ["<blank>"] + [chr(0x4E00 + i) for i in range(V - 2)] + ["<space>"]


def test_greedy_restatement_matches_reference_goldens():
    g = np.load(os.path.join(GOLD, "ctc_greedy_golden.npz"))
    V = int(g["vocab_size"])
    vocab = _vocab(V)
    for i in range(int(g["n_cases"])):
        score, text = dec.greedy_decoder(g[f"probs_{i}"], vocab)
        assert text == str(g[f"text_{i}"]), i
        # the reference sums np.float32 values with Python's sum(): float32 accumulation under
        # numpy>=2 (NEP 50), float64 under numpy 1.x.  Scores agree to float32 round-off.
        assert abs(score - float(g[f"score_{i}"])) <= 2e-5 * max(1.0, abs(score)), i
    texts = dec.greedy_decoder_batch([g[f"probs_{i}"] for i in g["batch_ids"]], vocab)
    assert texts == [str(t) for t in g["batch_texts"]]


def test_greedy_chunk_restatement_matches_reference_goldens():
    g = np.load(os.path.join(GOLD, "ctc_greedy_golden.npz"))
    vocab = _vocab(int(g["vocab_size"]))
    p = g[f"probs_{int(g['chunk_case'])}"]
    l1 = l2 = None
    for n, s in enumerate(range(0, p.shape[0], 16)):
        score, text, l1, l2 = dec.greedy_decoder_chunk(p[s:s + 16], vocab, l1, l2)
        assert text == str(g["chunk_texts"][n])
        assert abs(score - float(g["chunk_scores"][n])) <= 2e-5 * max(1.0, abs(score))


def _tiny():
    V, L = 64, 2
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=77, perturb_norm=True)
    x, lens = synth_features(2, 99, lens=[99, 61], seed=78)
    return ConformerOracle(sd, num_blocks=L), x, lens


def test_conformer_oracle_matches_committed_goldens():
    torch.set_num_threads(1)
    g = np.load(os.path.join(GOLD, "conformer_oracle_golden.npz"))
    o, x, lens = _tiny()
    probs, logits = o.get_encoder_out(x, lens, return_logits=True)
    assert np.abs(logits.numpy() - g["logits"]).max() <= 1e-4 * np.abs(g["logits"]).max()
    assert (probs.argmax(-1).numpy() == g["probs_argmax"]).mean() > 0.99
    p1, att, cnn = o.get_encoder_out_chunk(x[:1, :67], 0, -1)
    p2, att2, cnn2 = o.get_encoder_out_chunk(x[:1, 64:99], 16, -1, att, cnn)
    assert np.allclose(p1.numpy(), g["chunk1_probs"], atol=1e-5)
    assert np.allclose(p2.numpy(), g["chunk2_probs"], atol=1e-5)
    assert list(att2.shape) == list(g["att_cache_shape"])
    assert np.allclose(cnn2.numpy(), g["cnn_cache"], atol=1e-4)


def test_single_chunk_equals_full_utterance():
    """predict() on a streaming model = one giant chunk with empty caches
    and no mask == get_encoder_out for an un-padded single utterance (inference_predictor.py:127-137)."""
    o, x, lens = _tiny()
    full = o.get_encoder_out(x[:1], lens[:1])
    chunk, att, cnn = o.get_encoder_out_chunk(x[:1], 0, -1)
    assert np.allclose(full.numpy(), chunk.numpy(), atol=1e-6)
    assert att.shape == (2, 4, full.shape[1], 128) and cnn.shape == (2, 1, 256, 14)


def test_masked_frames_do_not_influence_valid_frames():
    """Keys beyond the utterance length are masked (attention.py:112-118, key j masked iff 4j >= len,
    subsampling.py:115) and PAD frames are zeroed around the causal conv module
    (convolution.py:104-106,138-140): garbage in feature rows that only feed masked frames
    (output frame i reads input rows 4i..4i+6) must leave the valid output frames bit-identical."""
    o, x, lens = _tiny()
    n_valid = (int(lens[1]) + 3) // 4
    base = o.get_encoder_out(x, lens).numpy()[1]
    x2 = x.copy()
    x2[1, 4 * (n_valid - 1) + 7:] = 123.0
    pert = o.get_encoder_out(x2, lens).numpy()[1]
    assert np.array_equal(base[:n_valid], pert[:n_valid])
    assert not np.array_equal(base[n_valid:], pert[n_valid:])
