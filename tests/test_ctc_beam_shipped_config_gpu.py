"""The beam search at the decode configuration every YAML of the reference SHIPS (configs/conformer.yml:78-92,
decoders/beam_search_decoder.py:9-42): beam_size 300, cutoff_prob 0.99, cutoff_top_n 40, a character n-gram scorer with
alpha 2.2 / beta 4.3 -- at full size (V = 4233 characters, T = 249 frames = a 10 s utterance, B = 8), against the C
oracle (oracle/ctc_beam_search_oracle.c; parity UNPINNED: paddlespeech_ctcdecoders / KenLM are third-party and absent,
see the oracle header), token sequences bit for bit.  Plus the wrappers' own default `cutoff_prob = 1.0`
(swig_wrapper.py:38,71): upstream then keeps EVERY character of every frame -- so does the kernel (no candidate cap)."""
import numpy as np
import pytest
import torch

from lm_util import read_arpa, write_synthetic_arpa
from test_ctc_beam_gpu import _oracle, _oracle_decode, _probs
from test_ctc_beam_lm_gpu import _oracle_lm_decode, _vocab

pytestmark = pytest.mark.gpu

V, T, BEAM, ALPHA, BETA = 4233, 249, 300, 2.2, 4.3


def _trained_like(rng, T, V):
    """Posteriors of the kind a trained CTC model emits: blank takes most frames, a character spike lasts 1 - 3 frames,
    a handful of confusable characters share the rest; a few frames are genuinely ambiguous."""
    logits = rng.standard_normal((T, V)).astype(np.float32)
    t = 0
    while t < T:
        if rng.random() < 0.55:
            n = int(rng.integers(1, 6))
            logits[t:t + n, 0] += 15.0
        else:
            n = int(rng.integers(1, 4))
            c = int(rng.integers(1, V))
            logits[t:t + n, c] += 14.0
            for alt in rng.integers(1, V, size=3):
                logits[t:t + n, alt] += float(rng.uniform(8.0, 13.0))
            if rng.random() < 0.5:
                logits[t:t + n, 0] += 12.0
        t += n
    e = np.exp(logits - logits.max(1, keepdims=True))
    return (e / e.sum(1, keepdims=True)).astype(np.float32)


def _table(rng, kind):
    return _trained_like(rng, T, V) if kind == "trained" else _probs(rng, T, V, kind)


@pytest.fixture(scope="module")
def big_lm(tmp_path_factory):
    """character 3-gram over ~80 % of the vocabulary (the rest is OOV to the scorer), ~60 k n-grams"""
    vocab = _vocab(V)
    rng = np.random.Generator(np.random.PCG64(99))
    known = [c for c in vocab[2:-1] if rng.random() < 0.8]
    path = str(tmp_path_factory.mktemp("lm") / "zh_char.arpa")
    write_synthetic_arpa(path, known, order=3, n_sent=3000, sent_len=24, seed=17)
    return vocab, path, read_arpa(path, vocab)


@pytest.mark.parametrize("kind", ["peaky", "flat", "trained"])
def test_shipped_configuration_with_scorer_full_size(big_lm, kind):
    from ppasr_amd.decoders.beam_search_decoder import Scorer, beam_search_ids
    vocab, arpa, lm = big_lm
    lib = _oracle()
    scorer = Scorer(ALPHA, BETA, arpa, vocab)
    B = 8
    rng = np.random.Generator(np.random.PCG64({"peaky": 1, "flat": 2, "trained": 3}[kind]))
    batch = np.stack([_table(rng, kind) for _ in range(B)])
    lens = np.array([T, T, T - 1, T - 40, T // 2, T, 17, T], np.int32)
    tokens, ln, sc, _ = beam_search_ids(torch.from_numpy(batch).cuda(), BEAM, 0.99, 40, 0, nbest=3, frame_lens=lens,
                                        ext_scorer=scorer)
    torch.cuda.synchronize()
    tokens, ln, sc = tokens.cpu().numpy(), ln.cpu().numpy(), sc.cpu().numpy()
    n_tok = 0
    for b in range(B):
        ref = _oracle_lm_decode(lib, [batch[b, :lens[b]]], V, BEAM, 0.99, 40, lm, ALPHA, BETA, 3)
        assert tokens[b, 0, :ln[b, 0]].tolist() == ref[0][0], (kind, b)
        assert abs(sc[b, 0] - ref[0][1]) <= 2e-4 * max(1.0, abs(ref[0][1])), (kind, b, sc[b, 0], ref[0][1])
        for r in range(1, len(ref)):  # the rest of the n-best list unless two scores sit within float noise
            if abs(ref[r][1] - ref[r - 1][1]) > 1e-3 and (r + 1 >= len(ref) or abs(ref[r + 1][1] - ref[r][1]) > 1e-3):
                assert tokens[b, r, :ln[b, r]].tolist() == ref[r][0], (kind, b, r)
        n_tok += len(ref[0][0])
    print(f"shipped config + scorer [{kind}]: {B} utterances, {n_tok} tokens identical to the oracle")
    assert n_tok > 0


@pytest.mark.parametrize("kind", ["peaky", "flat", "trained"])
def test_shipped_configuration_without_scorer_full_size(kind):
    from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
    lib = _oracle()
    B = 8
    rng = np.random.Generator(np.random.PCG64({"peaky": 11, "flat": 12, "trained": 13}[kind]))
    batch = np.stack([_table(rng, kind) for _ in range(B)])
    tokens, ln, sc, _ = beam_search_ids(torch.from_numpy(batch).cuda(), BEAM, 0.99, 40, 0, nbest=1)
    torch.cuda.synchronize()
    tokens, ln, sc = tokens.cpu().numpy(), ln.cpu().numpy(), sc.cpu().numpy()
    for b in range(B):
        ref = _oracle_decode(lib, batch[b], BEAM, 0.99, 40, 0, 1)
        assert tokens[b, 0, :ln[b, 0]].tolist() == ref[0][0], (kind, b)
        assert abs(sc[b, 0] - ref[0][1]) <= 1e-4 * max(1.0, abs(ref[0][1]))


@pytest.mark.parametrize("beam,top_n,kind,with_lm,Tn,B", [
    (10, 40, "peaky", False, 48, 3),   # swig_wrapper.py:38 defaults: cutoff_prob = 1.0 -> cutoff_top_n is ignored upstream
    (10, 40, "flat", False, 48, 3),
    (300, 40, "trained", False, 16, 2),  # (the oracle's trie creates and frees 1.27 M nodes per frame here: ~1 s per frame)
    (300, 40, "flat", False, 12, 1),
    (25, 40, "flat", True, 40, 2),
    (10, 5000, "flat", False, 48, 3),  # cutoff_top_n >= V as well: vocabulary order, nothing sorted
])
def test_unpruned_search_keeps_the_whole_vocabulary(big_lm, beam, top_n, kind, with_lm, Tn, B):
    """cutoff_prob = 1.0, V = 4233: all 4233 characters of every frame are candidates (beam x 4234 elements per frame:
    the element list lives in HBM when it does not fit LDS).  Equal to the oracle's genuinely unpruned decode."""
    from ppasr_amd.decoders.beam_search_decoder import Scorer, beam_search_ids
    vocab, arpa, lm = big_lm
    lib = _oracle()
    scorer = Scorer(ALPHA, BETA, arpa, vocab) if with_lm else None
    rng = np.random.Generator(np.random.PCG64(beam + top_n))
    batch = np.stack([_table(rng, kind)[:Tn] for _ in range(B)])
    tokens, ln, sc, _ = beam_search_ids(torch.from_numpy(batch).cuda(), beam, 1.0, top_n, 0, nbest=1, ext_scorer=scorer)
    torch.cuda.synchronize()
    for b in range(B):
        ref = _oracle_lm_decode(lib, [batch[b]], V, beam, 1.0, top_n, lm if with_lm else None, ALPHA, BETA, 1)
        got = tokens[b, 0, :int(ln[b, 0])].cpu().tolist()
        assert got == ref[0][0], (b, got, ref[0][0])
        assert abs(float(sc[b, 0]) - ref[0][1]) <= 2e-4 * max(1.0, abs(ref[0][1]))


@pytest.mark.parametrize("top_n,cutoff_prob", [(200, 0.9999), (1000, 0.999999)])
def test_wide_pruning_beyond_128_candidates(top_n, cutoff_prob):
    """cutoff_top_n above the old 128-candidate capacity with cutoff_prob < 1: the sorted list is cut by the cumulative
    probability or at top_n, whichever comes first."""
    from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
    lib = _oracle()
    rng = np.random.Generator(np.random.PCG64(top_n))
    batch = np.stack([_probs(rng, 40, V, "flat") for _ in range(2)])
    tokens, ln, sc, _ = beam_search_ids(torch.from_numpy(batch).cuda(), 20, cutoff_prob, top_n, 0, nbest=1)
    for b in range(2):
        ref = _oracle_decode(lib, batch[b], 20, cutoff_prob, top_n, 0, 1)
        assert tokens[b, 0, :int(ln[b, 0])].cpu().tolist() == ref[0][0], b
