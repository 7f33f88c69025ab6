"""CPU: KenLM binary (.klm) reader (ppasr_amd/csrc/klm.hip) -- every n-gram of a synthetic ARPA model written as a
probing / rest-probing / trie binary (tests/klm_writer.py) scores EXACTLY like the ARPA model itself under
Scorer::get_log_cond_prob, on 10 000 random contexts per model; format sniffing; refusal of the variants not read."""
import ctypes
import os

import numpy as np
import pytest

from klm_writer import patch_model_type, write_klm
from lm_util import write_synthetic_arpa
from ppasr_amd import _lib

CHARS = [chr(0x4E00 + i) for i in range(40)]
VOCAB = ["<blank>", "<unk>"] + CHARS + [chr(0x5000 + i) for i in range(8)] + ["<eos>"]  # 8 characters the LM does not know


def _load(path):
    lib = _lib.load()
    words = (ctypes.c_char_p * len(VOCAB))(*[w.encode("utf-8") for w in VOCAB])
    h = ctypes.c_void_p()
    _lib.check(lib.ppasr_lm_debug_load_host(str(path).encode(), words, len(VOCAB), ctypes.byref(h)))
    return lib, h


def _scores(lib, h, tokens_windows):
    """windows of acoustic token ids (bos = -1, eos = -2) -> get_log_cond_prob via the model's own word indices."""
    order = lib.ppasr_lm_order(h)
    bos, eos = lib.ppasr_lm_bos(h), lib.ppasr_lm_eos(h)
    out = np.empty(len(tokens_windows))
    win = (ctypes.c_int32 * order)()
    for i, w in enumerate(tokens_windows):
        for j, t in enumerate(w):
            win[j] = bos if t == -1 else (eos if t == -2 else lib.ppasr_lm_word_index(h, int(t)))
        out[i] = lib.ppasr_lm_debug_host_score(h, win)
    return out


def _windows(order, n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    wins = rng.integers(2, len(VOCAB) - 1, size=(n, order))
    # sentence starts (<s>-padded windows, as Scorer::make_ngram builds them) and sentence ends
    for i in range(0, n, 7):
        k = int(rng.integers(1, order))
        wins[i, :k] = -1
    wins[::11, -1] = -2
    return wins


@pytest.mark.parametrize("order,model_type", [(2, "probing"), (3, "probing"), (5, "probing"), (4, "rest_probing"),
                                               (2, "trie"), (3, "trie"), (5, "trie"),
                                               (2, "quant_trie"), (3, "quant_trie"), (5, "quant_trie"),
                                               (3, "array_trie"), (4, "array_trie"), (5, "array_trie"),
                                               (3, "quant_array_trie"), (5, "quant_array_trie")])
def test_klm_scores_equal_arpa_scores(tmp_path, order, model_type):
    arpa = write_synthetic_arpa(str(tmp_path / "lm.arpa"), CHARS, order=order, n_sent=300, sent_len=14, seed=order)
    klm = write_klm(arpa, str(tmp_path / "lm.klm"), model_type=model_type, multiplier=1.5 if order != 5 else 2.0,
                    prob_bits=13, backoff_bits=12, bhiksha_bits=64 if order != 4 else 3)
    lib, ha = _load(arpa)
    _, hk = _load(klm)
    try:
        assert lib.ppasr_lm_format(ha) == b"arpa"
        assert lib.ppasr_lm_format(hk) == ("klm-" + model_type.replace("_", "-")).encode()
        assert lib.ppasr_lm_order(hk) == order == lib.ppasr_lm_order(ha)
        assert lib.ppasr_lm_ngram_count(hk) == lib.ppasr_lm_ngram_count(ha)
        assert lib.ppasr_lm_is_character_based(hk) == 1
        wins = _windows(order, 10000, 100 + order)
        sa, sk = _scores(lib, ha, wins), _scores(lib, hk, wins)
        assert np.array_equal(sa, sk)                       # identical floats, not merely close
        assert (sa == -1000.0).any() and (sa > -1000.0).sum() > 2500   # OOV windows and scored windows both occur
        assert len(np.unique(sa)) > 200                     # back-off paths of every depth are exercised
    finally:
        lib.ppasr_lm_destroy(ha)
        lib.ppasr_lm_destroy(hk)


def test_refusals_and_corruption(tmp_path):
    arpa = write_synthetic_arpa(str(tmp_path / "lm.arpa"), CHARS, order=3, seed=9)
    klm = write_klm(arpa, str(tmp_path / "lm.klm"), "probing")
    lib = _lib.load()
    words = (ctypes.c_char_p * len(VOCAB))(*[w.encode("utf-8") for w in VOCAB])
    h = ctypes.c_void_p()

    def status(path):
        return lib.ppasr_lm_debug_load_host(str(path).encode(), words, len(VOCAB), ctypes.byref(h))

    for mt in (6, 17):     # no such KenLM model type
        patch_model_type(klm, mt)
        assert status(klm) == _lib.PPASR_EUNSUPPORTED, lib.ppasr_last_error()
    for mt in (3, 4, 5):   # a probing file relabelled as a (quantised / array) trie: the layout checks must catch it
        patch_model_type(klm, mt)
        assert status(klm) in (_lib.PPASR_EINVAL, _lib.PPASR_EUNSUPPORTED), lib.ppasr_last_error()
    patch_model_type(klm, 0)
    assert status(klm) == 0
    lib.ppasr_lm_destroy(h)
    data = open(klm, "rb").read()
    open(klm, "wb").write(data[:-5])                      # vocabulary strings cut short
    assert status(klm) == _lib.PPASR_EINVAL
    open(klm, "wb").write(data + b"xyz\x00")              # trailing bytes the layout does not explain
    assert status(klm) == _lib.PPASR_EINVAL
    open(klm, "wb").write(data[:60] + b"\x01" + data[61:])  # sanity block
    assert status(klm) == _lib.PPASR_EINVAL
    open(klm, "wb").write(data.replace(b"version 5", b"version 4", 1))
    assert status(klm) == _lib.PPASR_EUNSUPPORTED
    open(klm, "wb").write(data[:100] + b"\x00" + data[101:])  # has_vocabulary = false
    assert status(klm) == _lib.PPASR_EUNSUPPORTED
    # a word-based model (words of more than one character) needs the space token in the acoustic vocabulary
    warpa = str(tmp_path / "w.arpa")
    write_synthetic_arpa(warpa, ["你好", "世界", "今", "天"], order=2, seed=3)
    assert status(warpa) == _lib.PPASR_EINVAL and b"space" in lib.ppasr_last_error()
    assert status(write_klm(warpa, str(tmp_path / "w.klm"), "trie")) == _lib.PPASR_EINVAL


def test_word_based_model_builds_its_dictionary(tmp_path):
    """Scorer::fill_dictionary: every LM word that can be spelt with the acoustic characters goes into the dictionary
    (get_dict_size); the model is word-based as soon as one word has more than one character; ARPA and every .klm type
    agree."""
    lib = _lib.load()
    letters = list("abcdefghijklmnopqrstuvwxyz'")
    vocab = ["<blank>"] + letters + ["<space>", "<eos>"]
    lm_words = ["the", "cat", "sat", "on", "a", "mat", "it's", "naïve", "x-ray", "zebra"]   # 2 cannot be spelt
    arpa = write_synthetic_arpa(str(tmp_path / "w.arpa"), lm_words, order=3, n_sent=200, sent_len=8, seed=4)
    words = (ctypes.c_char_p * len(vocab))(*[w.encode("utf-8") for w in vocab])
    sizes = set()
    for path in [arpa] + [write_klm(arpa, str(tmp_path / f"w_{mt}.klm"), mt) for mt in
                          ("probing", "trie", "quant_trie", "array_trie", "quant_array_trie")]:
        h = ctypes.c_void_p()
        _lib.check(lib.ppasr_lm_debug_load_host(str(path).encode(), words, len(vocab), ctypes.byref(h)))
        assert lib.ppasr_lm_is_character_based(h) == 0
        assert lib.ppasr_lm_space_id(h) == vocab.index("<space>")
        sizes.add(lib.ppasr_lm_dict_size(h))
        lib.ppasr_lm_destroy(h)
    assert sizes == {8}    # "naïve" and "x-ray" contain characters outside the acoustic vocabulary
