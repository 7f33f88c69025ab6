"""GPU: beam search with a WORD-based external scorer (configs/english_example.yml style: letters + space, a word n-gram
model) against the C oracle's restatement of upstream's dictionary-constrained search (parity UNPINNED like the rest of
the beam search: paddlespeech_ctcdecoders is not in the tree).  The oracle gets its dictionary from THIS file (a
character trie built in Python), the library builds its own from the model's vocabulary (csrc/lm.hip): agreement checks
both.  Covered: ARPA and .klm word models, the reset-after-a-word quirk of PathTrie::get_path_trie, the final
"score the last word" step, chunked == one-shot."""
import ctypes

import numpy as np
import pytest
import torch

from klm_writer import write_klm
from lm_util import read_arpa, write_synthetic_arpa
from test_ctc_beam_gpu import _oracle
from test_ctc_beam_lm_gpu import _ptr

pytestmark = pytest.mark.gpu

LETTERS = list("abcdefghijklmnopqrstuvwxyz'")
VOCAB = ["<blank>"] + LETTERS + ["<space>", "<eos>"]
SPACE = VOCAB.index("<space>")
WORDS = ["the", "cat", "sat", "on", "a", "mat", "it's", "at", "an", "ant", "anthem", "then", "there", "here", "he", "her",
         "hat", "that", "this", "is", "in", "inn", "to", "too", "tom", "cats", "sit", "sits", "so", "soon", "no", "not",
         "note", "notes", "one", "once", "we", "were", "where", "when", "what", "who", "why", "how", "now", "new", "news"]


def _dictionary(lm, vocab=None):
    """character trie of every LM word + space in CSR form, nodes numbered by a BFS of my own (independent of the
    library's builder): -> (first, arc_char, arc_next, word)"""
    tok = {c: i for i, c in enumerate(vocab or VOCAB)}
    nodes = [{}]
    word_at = {}
    for w, wid in lm["words"].items():
        if w in ("<unk>", "<s>", "</s>") or any(ch not in tok for ch in w):
            continue
        s = 0
        for ch in list(w) + ["<space>"]:
            c = tok[ch]
            if c not in nodes[s]:
                nodes[s][c] = len(nodes)
                nodes.append({})
            s = nodes[s][c]
        word_at[s] = wid
    first, arc_char, arc_next, word = [0], [], [], []
    for n, arcs in enumerate(nodes):
        for c in sorted(arcs):
            arc_char.append(c)
            arc_next.append(arcs[c])
        first.append(len(arc_char))
        word.append(word_at.get(n, 0))
    as32 = lambda a: np.asarray(a, np.int32)
    return as32(first), as32(arc_char), as32(arc_next), as32(word)


def _spoken_probs(rng, sentence, V, noise=0.12, blank_p=0.45):
    """A frame table that 'speaks' the sentence: per character 1-3 frames peaked on it, blanks in between, plus noise on
    confusable characters -- so that the beam holds many spellings the dictionary has to sort out."""
    tok = {c: i for i, c in enumerate(VOCAB)}
    ids = []
    for wi, w in enumerate(sentence):
        ids += [tok[ch] for ch in w] + [SPACE]
    rows = []
    for c in ids:
        for _ in range(int(rng.integers(2, 4))):
            p = rng.random(V) ** 6 * noise
            p[c] += 1.0
            p[0] += blank_p * rng.random()
            rows.append(p)
        if rng.random() < 0.5:
            p = rng.random(V) ** 6 * noise
            p[0] += 1.0
            rows.append(p)
    t = np.asarray(rows, np.float64)
    t /= t.sum(-1, keepdims=True)
    return t.astype(np.float32)


def _oracle_word_decode(lib, chunks, V, beam, cutoff_prob, top_n, lm, dic, alpha, beta, nbest):
    lib.ctc_beam_oracle_set_lm.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5 + [
        ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    lib.ctc_beam_oracle_set_dictionary.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
    h = lib.ctc_beam_oracle_create(V, beam, ctypes.c_double(cutoff_prob), top_n, 0)
    lib.ctc_beam_oracle_set_lm(h, lm["order"], len(lm["gram_n"]), _ptr(lm["gram_n"]), _ptr(lm["gram_w"]), _ptr(lm["prob"]),
                               _ptr(lm["backoff"]), _ptr(lm["tok2lm"]), lm["bos"], lm["eos"], alpha, beta)
    first, arc_char, arc_next, word = dic
    lib.ctc_beam_oracle_set_dictionary(h, SPACE, len(word), _ptr(first), _ptr(arc_char), _ptr(arc_next), _ptr(word))
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


@pytest.mark.parametrize("fmt,beam,order,alpha,beta", [("arpa", 20, 3, 1.9, 0.3),      # english_example.yml weights
                                                        ("arpa", 100, 2, 1.9, 0.3),
                                                        ("arpa", 160, 3, 1.9, 0.3),     # (the 768-thread form of the search)
                                                        ("arpa", 8, 3, 0.8, -0.5),
                                                        ("trie", 30, 3, 1.9, 0.3),
                                                        ("probing", 30, 3, 1.9, 0.3),
                                                        ("quant_array_trie", 30, 3, 1.9, 0.3)])
def test_word_based_scorer_matches_c_oracle(tmp_path, fmt, beam, order, alpha, beta):
    from ppasr_amd.decoders.beam_search_decoder import Scorer, beam_search_ids
    lib = _oracle()
    V = len(VOCAB)
    rng = np.random.Generator(np.random.PCG64(beam * 7 + order))
    arpa = write_synthetic_arpa(str(tmp_path / "w.arpa"), WORDS, order=order, n_sent=300, sent_len=8, seed=order)
    lm = read_arpa(arpa, VOCAB)
    path = arpa if fmt == "arpa" else write_klm(arpa, str(tmp_path / "w.klm"), fmt)
    with pytest.warns(RuntimeWarning) if fmt != "arpa" and not Scorer._klm_warned else _nullcontext():
        scorer = Scorer(alpha, beta, path, VOCAB)
    assert not scorer.is_character_based() and scorer.get_max_order() == order and scorer.get_dict_size() == len(WORDS)
    dic = _dictionary(lm)
    B = 4
    sents = [[WORDS[int(i)] for i in rng.integers(0, len(WORDS), size=int(rng.integers(2, 7)))] for _ in range(B)]
    tabs = [_spoken_probs(rng, s, V) for s in sents]
    T = max(t.shape[0] for t in tabs)
    batch = np.zeros((B, T, V), np.float32)
    lens = np.array([t.shape[0] for t in tabs], np.int32)
    for b, t in enumerate(tabs):
        batch[b, :t.shape[0]] = t
    nbest = min(beam, 3)
    tokens, ln, scores, _ = beam_search_ids(torch.from_numpy(batch).cuda(), beam, 0.99, 40, 0, frame_lens=lens, nbest=nbest,
                                            ext_scorer=scorer)
    torch.cuda.synchronize()
    tokens, ln, scores = tokens.cpu().numpy(), ln.cpu().numpy(), scores.cpu().numpy()
    spoken = 0
    for b in range(B):
        ref = _oracle_word_decode(lib, [batch[b, :lens[b]]], V, beam, 0.99, 40, lm, dic, alpha, beta, nbest)
        got = tokens[b, 0, :ln[b, 0]].tolist()
        assert got == ref[0][0], (b, "".join(VOCAB[i] if i != SPACE else " " for i in got),
                                  "".join(VOCAB[i] if i != SPACE else " " for i in ref[0][0]))
        assert abs(scores[b, 0] - ref[0][1]) <= 2e-4 * max(1.0, abs(ref[0][1]))
        text = "".join(VOCAB[i] if i != SPACE else " " for i in got).split()
        assert all(w in WORDS for w in text[:-1])      # every completed word is a dictionary word
        spoken += int(text == sents[b])
    print(f"word LM [{fmt}, beam {beam}]: {spoken}/{B} tables decoded to exactly the sentence they speak")


def test_word_based_scorer_on_an_unpruned_wide_vocabulary(tmp_path):
    """cutoff_prob = 1.0 keeps the whole vocabulary (upstream sorts it when cutoff_top_n < V) -- more than 128 characters
    per frame is the wide form of the search (records and element lists in HBM scratch), here with the word-based scorer
    and its dictionary on top: 130 extra symbols that no word uses behind the letters."""
    from ppasr_amd.decoders.beam_search_decoder import Scorer, beam_search_ids
    lib = _oracle()
    vocab = VOCAB + [f"<x{i}>" for i in range(130)]
    V = len(vocab)
    rng = np.random.Generator(np.random.PCG64(99))
    arpa = write_synthetic_arpa(str(tmp_path / "w.arpa"), WORDS, order=3, n_sent=300, sent_len=8, seed=5)
    lm = read_arpa(arpa, vocab)
    scorer = Scorer(1.9, 0.3, arpa, vocab)
    assert not scorer.is_character_based()
    dic = _dictionary(lm, vocab)
    sents = [["the", "cat", "sat"], ["here", "we", "were", "soon"]]
    tabs = [_spoken_probs(rng, s, V, noise=0.05) for s in sents]
    T = max(t.shape[0] for t in tabs)
    batch = np.zeros((len(tabs), T, V), np.float32)
    lens = np.array([t.shape[0] for t in tabs], np.int32)
    for b, t in enumerate(tabs):
        batch[b, :t.shape[0]] = t
    beam = 12
    tokens, ln, scores, _ = beam_search_ids(torch.from_numpy(batch).cuda(), beam, 1.0, 40, 0, frame_lens=lens, nbest=2,
                                            ext_scorer=scorer)
    torch.cuda.synchronize()
    tokens, ln, scores = tokens.cpu().numpy(), ln.cpu().numpy(), scores.cpu().numpy()
    for b in range(len(tabs)):
        ref = _oracle_word_decode(lib, [batch[b, :lens[b]]], V, beam, 1.0, 40, lm, dic, 1.9, 0.3, 2)
        assert tokens[b, 0, :ln[b, 0]].tolist() == ref[0][0], b
        assert abs(scores[b, 0] - ref[0][1]) <= 2e-4 * max(1.0, abs(ref[0][1]))


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def test_word_based_chunked_equals_one_shot(tmp_path):
    from ppasr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    rng = np.random.Generator(np.random.PCG64(12))
    arpa = write_synthetic_arpa(str(tmp_path / "w.arpa"), WORDS, order=3, n_sent=300, sent_len=8, seed=2)
    dec = BeamSearchDecoder(1.9, 0.3, 30, 0.99, 40, VOCAB, language_model_path=arpa)
    p = _spoken_probs(rng, ["the", "cat", "sat", "on", "the", "mat"], len(VOCAB))
    off_score, off_text = dec.decode_beam_search_offline(p)
    assert off_text.replace("<space>", " ").split()[0] == "the"
    for lo in range(0, p.shape[0], 7):
        score, text = dec.decode_chunk(p[lo:lo + 7][None], [min(7, p.shape[0] - lo)])
    assert text == off_text and abs(score - off_score) <= 1e-4 * max(1.0, abs(off_score))
