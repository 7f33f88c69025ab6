"""ORACLE (test infrastructure, not product code) -- numpy restatement of PPASR's
CTC greedy decoders, ``ppasr/decoders/ctc_greedy_decoder.py``.

Pinned: ``tests/golden/ctc_greedy_golden.npz`` was produced by importing the
reference module itself (``tests/golden/make_goldens.py``); ``tests/test_oracle.py``
checks this restatement against those vectors bit-for-bit (token ids) and to
1 ulp of float64 (scores).

Returns token-id lists as well as text so integer parity can be asserted.
"""
import numpy as np


def greedy_tokens(probs_seq, blank_index=0):
    """ctc_greedy_decoder.py:21-25 -> (collapsed ids, per-frame argmax, non-blank max probs)."""
    probs_seq = np.asarray(probs_seq)
    max_index = probs_seq.argmax(axis=1)  # first max wins (:21)
    nonblank = max_index != blank_index
    max_prob = probs_seq[np.arange(len(max_index)), max_index][nonblank]  # :22
    keep = np.ones(len(max_index), bool)
    keep[1:] = max_index[1:] != max_index[:-1]  # groupby collapse (:24)
    ids = max_index[keep]
    ids = ids[ids != blank_index]  # :25
    return ids.astype(np.int64), max_index.astype(np.int64), max_prob


def _score(max_prob_list):
    # :28-30  Python-float (fp64) mean of the f32 values * 100
    if len(max_prob_list) == 0:
        return 0
    return float(sum(max_prob_list) / len(max_prob_list)) * 100.0


def greedy_decoder(probs_seq, vocabulary, blank_index=0):
    """ctc_greedy_decoder.py:6-31"""
    ids, _, max_prob = greedy_tokens(probs_seq, blank_index)
    text = "".join(vocabulary[i] for i in ids)
    return _score(list(max_prob)), text.replace("<space>", " ")


def greedy_decoder_batch(probs_split, vocabulary, blank_index=0):
    """ctc_greedy_decoder.py:34-49 (no length trimming: all T' rows are decoded)."""
    return [greedy_decoder(p, vocabulary, blank_index)[1] for p in probs_split]


def greedy_decoder_chunk(probs_seq, vocabulary, last_max_prob_list=None, last_max_index_list=None, blank_index=0):
    """ctc_greedy_decoder.py:52-89.  NB the reference's argument names are swapped:
    ``last_max_prob_list`` accumulates argmax *indices* and ``last_max_index_list``
    accumulates non-blank max *probabilities* (:78-79); kept as is."""
    if last_max_prob_list is None:
        last_max_prob_list = []
    if last_max_index_list is None:
        last_max_index_list = []
    _, max_index, max_prob = greedy_tokens(probs_seq, blank_index)
    last_max_prob_list.extend(list(max_index))
    last_max_index_list.extend(list(max_prob))
    hist = np.asarray(last_max_prob_list, np.int64)
    keep = np.ones(len(hist), bool)
    keep[1:] = hist[1:] != hist[:-1]
    ids = hist[keep]
    ids = ids[ids != blank_index]
    text = "".join(vocabulary[i] for i in ids)
    return _score(last_max_index_list), text.replace("<space>", " "), last_max_prob_list, last_max_index_list
