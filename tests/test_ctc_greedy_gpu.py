"""GPU parity of the HIP greedy decoder against vectors produced by the REFERENCE's own
ppasr/decoders/ctc_greedy_decoder.py (tests/golden/ctc_greedy_golden.npz) -- token ids / text
bit-exact, scores to float32 round-off (the reference accumulates np.float32 with sum())."""
import os

import numpy as np
import pytest
import torch

from oracle import ctc_decoders_oracle as dec

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ctc_greedy_golden.npz")


def _vocab(V):
    return ["<blank>"] + [chr(0x4E00 + i) for i in range(V - 2)] + ["<space>"]


def test_greedy_decoder_matches_reference_goldens():
    from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decoder, greedy_decoder_batch
    g = np.load(GOLD)
    vocab = _vocab(int(g["vocab_size"]))
    for i in range(int(g["n_cases"])):
        score, text = greedy_decoder(g[f"probs_{i}"], vocab)
        assert text == str(g[f"text_{i}"]), i
        assert abs(score - float(g[f"score_{i}"])) <= 2e-5 * max(1.0, abs(score)), i
    batch = np.stack([g[f"probs_{i}"] for i in g["batch_ids"]])
    assert greedy_decoder_batch(torch.from_numpy(batch).cuda(), vocab) == [str(t) for t in g["batch_texts"]]
    assert greedy_decoder_batch(list(batch), vocab) == [str(t) for t in g["batch_texts"]]


def test_greedy_decoder_chunk_matches_reference_goldens():
    from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decoder_chunk
    g = np.load(GOLD)
    vocab = _vocab(int(g["vocab_size"]))
    p = g[f"probs_{int(g['chunk_case'])}"]
    l1 = l2 = None
    for n, s in enumerate(range(0, p.shape[0], 16)):
        score, text, l1, l2 = greedy_decoder_chunk(p[s:s + 16], vocab, l1, l2)
        assert text == str(g["chunk_texts"][n])
        assert abs(score - float(g["chunk_scores"][n])) <= 2e-5 * max(1.0, abs(score))


@pytest.mark.parametrize("B,T,V", [(1, 1, 2), (3, 249, 4233), (32, 249, 4233), (2, 750, 4233), (5, 64, 100)])
def test_greedy_ids_match_oracle_large(B, T, V):
    """Bench-size tables and edge shapes vs the numpy oracle; ties (quantised probs) -> lowest index."""
    from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decode_ids
    rng = np.random.Generator(np.random.PCG64(B * 1000 + T))
    p = np.round(rng.random((B, T, V)) * 64).astype(np.float32) / 64  # many exact ties
    p[:, :, 0] += (rng.random((B, T)) < 0.5).astype(np.float32)       # ~half the frames blank
    lens = rng.integers(0, T + 1, size=B).astype(np.int32)
    for fl in (None, lens):
        tokens, n, score, fa, fp = greedy_decode_ids(torch.from_numpy(p).cuda(), fl)
        torch.cuda.synchronize()
        for b in range(B):
            nb = T if fl is None else int(lens[b])
            ids, max_index, max_prob = dec.greedy_tokens(p[b, :nb])
            assert np.array_equal(fa[b, :nb].cpu().numpy(), max_index)
            assert np.array_equal(tokens[b, :int(n[b])].cpu().numpy(), ids)
            assert (tokens[b, int(n[b]):] == -1).all()
            ref = float(np.mean(max_prob.astype(np.float64))) * 100 if len(max_prob) else 0.0
            assert abs(float(score[b]) - ref) <= 1e-9 * max(1.0, ref)
