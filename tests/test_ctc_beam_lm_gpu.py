"""GPU parity of the beam search WITH the external scorer (character-based n-gram LM, csrc/lm.h + ctc_beam.hip) against
the C oracle's restatement of the `ext_scorer` branch of paddlespeech_ctcdecoders (parity UNPINNED: third-party, not in
the tree; KenLM unavailable -- the LM is a synthetic ARPA file, parsed independently by tests/lm_util.py for the oracle)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from lm_util import read_arpa, write_synthetic_arpa
from test_ctc_beam_gpu import _oracle, _probs

pytestmark = pytest.mark.gpu


def _vocab(V):
    return ["<blank>", "<unk>"] + [chr(0x4E00 + i) for i in range(V - 3)] + ["<eos>"]


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _oracle_lm_decode(lib, chunks, V, beam, cutoff_prob, top_n, lm, alpha, beta, nbest):
    lib.ctc_beam_oracle_set_lm.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5 + [
        ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    h = lib.ctc_beam_oracle_create(V, beam, ctypes.c_double(cutoff_prob), top_n, 0)
    if lm is not None:
        lib.ctc_beam_oracle_set_lm(h, lm["order"], len(lm["gram_n"]), _ptr(lm["gram_n"]), _ptr(lm["gram_w"]),
                                   _ptr(lm["prob"]), _ptr(lm["backoff"]), _ptr(lm["tok2lm"]), lm["bos"], lm["eos"],
                                   alpha, beta)
    total = 0
    for c in chunks:
        c = np.ascontiguousarray(c, np.float32)
        lib.ctc_beam_oracle_next(h, _ptr(c), c.shape[0])
        total += c.shape[0]
    L = max(total, 1)
    tokens = np.empty((nbest, L), np.int32)
    lens = np.empty(nbest, np.int32)
    scores = np.empty(nbest, np.float64)
    n = lib.ctc_beam_oracle_result(h, nbest, L, _ptr(tokens), _ptr(lens), _ptr(scores))
    lib.ctc_beam_oracle_free(h)
    return [(tokens[i, :lens[i]].tolist(), scores[i]) for i in range(n)]


@pytest.mark.parametrize("T,V,beam,order,alpha,beta,kind", [
    (60, 120, 10, 3, 2.2, 4.3, "peaky"),     # PPASR's Mandarin defaults (conformer.yml:78-92), beam reduced
    (60, 120, 10, 3, 2.2, 4.3, "flat"),
    (40, 120, 30, 2, 1.9, 0.3, "peaky"),     # english_example.yml weights
    (80, 300, 8, 5, 0.5, -1.5, "peaky"),     # negative beta: max(0, beta) in min_cutoff
    (50, 4233, 10, 4, 2.2, 4.3, "peaky"),
    (30, 120, 300, 3, 2.2, 4.3, "flat"),     # the reference's default beam_size
    (20, 4233, 300, 3, 2.2, 4.3, "flat"),    # ... with the full vocabulary (LDS budget of beam x candidates)
    (1, 120, 5, 3, 2.2, 4.3, "flat"),
])
def test_beam_search_with_scorer_matches_c_oracle(tmp_path, T, V, beam, order, alpha, beta, kind):
    from ppasr_amd.decoders.beam_search_decoder import Scorer, beam_search_ids
    lib = _oracle()
    vocab = _vocab(V)
    rng = np.random.Generator(np.random.PCG64(T * 13 + V + beam + order))
    # the LM knows ~80 % of the characters: the rest are OOV (OOV_SCORE branch)
    known = [c for c in vocab[2:-1] if rng.random() < 0.8][:400]
    arpa = write_synthetic_arpa(str(tmp_path / "lm.arpa"), known, order=order, seed=order + beam)
    lm = read_arpa(arpa, vocab)
    scorer = Scorer(alpha, beta, arpa, vocab)
    assert scorer.get_max_order() == order and scorer.is_character_based()
    assert scorer.ngram_count() == len(lm["gram_n"])
    B = 3
    batch = np.stack([_probs(rng, T, V, kind) for _ in range(B)])
    nbest = min(beam, 4)
    tokens, lens, scores, _ = beam_search_ids(torch.from_numpy(batch).cuda(), beam, 0.99, 40, 0, nbest=nbest,
                                              ext_scorer=scorer)
    t0, l0, _, _ = beam_search_ids(torch.from_numpy(batch).cuda(), beam, 0.99, 40, 0, nbest=1)
    torch.cuda.synchronize()
    tokens, lens, scores = tokens.cpu().numpy(), lens.cpu().numpy(), scores.cpu().numpy()
    differs = 0
    for b in range(B):
        ref = _oracle_lm_decode(lib, [batch[b]], V, beam, 0.99, 40, lm, alpha, beta, nbest)
        assert tokens[b, 0, :lens[b, 0]].tolist() == ref[0][0], (b, ref[0])
        assert abs(scores[b, 0] - ref[0][1]) <= 2e-4 * max(1.0, abs(ref[0][1]))
        for r in range(1, len(ref)):
            if abs(ref[r][1] - ref[r - 1][1]) > 1e-3 and (r + 1 >= len(ref) or abs(ref[r + 1][1] - ref[r][1]) > 1e-3):
                assert tokens[b, r, :lens[b, r]].tolist() == ref[r][0], (b, r)
        differs += tokens[b, 0, :lens[b, 0]].tolist() != t0[b, 0, :int(l0[b, 0])].cpu().numpy().tolist()
    if T >= 40 and kind == "flat":
        assert differs > 0  # the scorer must actually steer the search on ambiguous posteriors


def test_scorer_streaming_and_decoder_object(tmp_path):
    """BeamSearchDecoder(language_model_path=...) builds the Scorer (beam_search_decoder.py:28-29); chunked decoding with
    the scorer equals the one-shot decode and the oracle object fed the same chunks."""
    from ppasr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    lib = _oracle()
    V, beam = 200, 10
    vocab = _vocab(V)
    rng = np.random.Generator(np.random.PCG64(11))
    arpa = write_synthetic_arpa(str(tmp_path / "lm.arpa"), vocab[2:150], order=3, seed=4)
    lm = read_arpa(arpa, vocab)
    p = _probs(rng, 96, V, "flat")
    dec = BeamSearchDecoder(2.2, 4.3, beam, 0.99, 40, vocab, language_model_path=arpa)
    off_score, off_text = dec.decode_beam_search_offline(p)
    ref = _oracle_lm_decode(lib, [p], V, beam, 0.99, 40, lm, 2.2, 4.3, 1)
    assert off_text == "".join(vocab[i] for i in ref[0][0])
    assert abs(off_score - ref[0][1]) <= 2e-4 * max(1.0, abs(ref[0][1]))
    text = None
    for i in range(0, 96, 16):
        _, text = dec.decode_chunk(p[None, i:i + 16], [16])
    assert text == off_text
    dec.reset_decoder()
    assert dec.decode_batch_beam_search_offline([p, p]) == [off_text, off_text]


def test_scorer_rejects_unsupported_models(tmp_path):
    from ppasr_amd import _lib
    from ppasr_amd.decoders.beam_search_decoder import Scorer
    vocab = _vocab(50)
    klm = tmp_path / "x.klm"   # a KenLM magic followed by garbage: refused, not mis-read
    klm.write_bytes(b"mmap lm http://kheafield.com/code format version 5\n\x00" + bytes(200))
    with pytest.raises(_lib.PPASRHipError, match="sanity block"):
        Scorer(1.0, 1.0, str(klm), vocab)
    word = tmp_path / "w.arpa"
    word.write_text("\\data\\\nngram 1=4\n\n\\1-grams:\n-1.0\t<unk>\n-99\t<s>\t-0.5\n-1.2\t</s>\n-2.0\thello\t-0.3\n\n\\end\\\n")
    with pytest.raises(_lib.PPASRHipError, match="word-based"):
        Scorer(1.0, 1.0, str(word), vocab)
    with pytest.raises(Exception, match="not found"):
        Scorer(1.0, 1.0, str(tmp_path / "missing.arpa"), vocab)


def test_seeded_fuzz_against_oracle(tmp_path):
    """Random (frames, vocabulary, beam, pruning, scorer on/off, weights) draws: best hypothesis and score equal the C oracle."""
    from ppasr_amd.decoders.beam_search_decoder import Scorer, beam_search_ids
    lib = _oracle()
    for seed in range(24):
        rng = np.random.Generator(np.random.PCG64(5000 + seed))
        T = int(rng.integers(1, 100))
        V = int(rng.choice([40, 97, 300, 1000, 4233]))
        beam = int(rng.choice([1, 2, 5, 10, 33, 100, 300]))
        order = int(rng.integers(2, 6))
        use_lm = bool(rng.integers(0, 2))
        kind = str(rng.choice(["peaky", "flat"]))
        cp, tn = float(rng.choice([0.99, 0.9, 0.5])), int(rng.choice([40, 10, 3]))
        alpha, beta = float(rng.uniform(0.2, 3)), float(rng.uniform(-2, 5))
        vocab = _vocab(V)
        lm = scorer = None
        if use_lm:
            known = [c for c in vocab[2:-1] if rng.random() < 0.8][:300]
            arpa = write_synthetic_arpa(str(tmp_path / f"lm{seed}.arpa"), known, order=order, seed=seed)
            lm = read_arpa(arpa, vocab)
            scorer = Scorer(alpha, beta, arpa, vocab)
        p = _probs(rng, T, V, kind)
        tk, ln, sc, _ = beam_search_ids(torch.from_numpy(p)[None].cuda(), beam, cp, tn, 0, nbest=1, ext_scorer=scorer)
        ref = _oracle_lm_decode(lib, [p], V, beam, cp, tn, lm, alpha, beta, 1)
        got = tk[0, 0, :int(ln[0, 0])].cpu().numpy().tolist()
        assert got == ref[0][0], (seed, T, V, beam, use_lm)
        assert abs(float(sc[0, 0]) - ref[0][1]) <= 2e-4 * max(1.0, abs(ref[0][1])), (seed, float(sc[0, 0]), ref[0][1])


@pytest.mark.parametrize("model_type,order", [("probing", 3), ("trie", 4), ("rest_probing", 2)])
def test_scorer_from_klm_binary_equals_scorer_from_arpa(tmp_path, model_type, order):
    """`Scorer(alpha, beta, 'lm.klm', vocab)` -- the file type PPASR ships (decoders/beam_search_decoder.py:19-29): the
    KenLM binary (written by tests/klm_writer.py from the same ARPA model) gives the SAME hypotheses and scores as the
    ARPA model, i.e. the device-side KenLM key chain (probing) / trie walk (trie) address the same n-grams."""
    from klm_writer import write_klm
    from ppasr_amd.decoders.beam_search_decoder import Scorer, beam_search_ids
    T, V, beam, alpha, beta = 70, 300, 20, 2.2, 4.3
    vocab = _vocab(V)
    rng = np.random.Generator(np.random.PCG64(77 + order))
    known = [c for c in vocab[2:-1] if rng.random() < 0.8]
    arpa = write_synthetic_arpa(str(tmp_path / "lm.arpa"), known, order=order, n_sent=400, seed=order)
    klm = write_klm(arpa, str(tmp_path / "zh.klm"), model_type=model_type)
    sa, sk = Scorer(alpha, beta, arpa, vocab), Scorer(alpha, beta, klm, vocab)
    assert sk.get_max_order() == order and sk.is_character_based() and sk.ngram_count() == sa.ngram_count()
    batch = torch.from_numpy(np.stack([_probs(rng, T, V, k) for k in ("peaky", "flat", "peaky")])).cuda()
    ta, la, ca, _ = beam_search_ids(batch, beam, 0.99, 40, 0, nbest=4, ext_scorer=sa)
    tk, lk, ck, _ = beam_search_ids(batch, beam, 0.99, 40, 0, nbest=4, ext_scorer=sk)
    torch.cuda.synchronize()
    assert torch.equal(la, lk) and torch.equal(ta, tk) and torch.equal(ca, ck)
    assert int(la[:, 0].min()) > 0
