"""The one-wave-per-utterance beam kernel (k_ctc_beam_wave: beam <= 16, <= 64 pruned characters per frame) against the
block-wide kernel (k_ctc_beam) it replaces for small beams: same arithmetic, keys and element order, so the two must
return IDENTICAL beams -- tokens, lengths and scores bit for bit -- including on inputs built to produce exact score
ties at the cut, with and without the external scorer, one-shot and chunk by chunk.  The wave kernel is selected with
PPASR_BEAM_WAVE=1 (read at every launch); both routes are also held to the C oracle here."""
import os

import numpy as np
import pytest
import torch

from lm_util import write_synthetic_arpa
from test_ctc_beam_gpu import _oracle, _oracle_decode, _probs

pytestmark = pytest.mark.gpu


class _route:
    def __init__(self, wave):
        self.wave = wave

    def __enter__(self):
        self.old = os.environ.get("PPASR_BEAM_WAVE")
        os.environ["PPASR_BEAM_WAVE"] = "1" if self.wave else "0"

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("PPASR_BEAM_WAVE", None)
        else:
            os.environ["PPASR_BEAM_WAVE"] = self.old


def _tie_probs(rng, T, V, levels=4):
    """Probabilities drawn from a handful of exactly representable values: many hypotheses share a score exactly."""
    vals = np.array([2.0 ** -(2 + i) for i in range(levels)], np.float32)
    p = vals[rng.integers(0, levels, size=(T, V))].astype(np.float32)
    p[:, 0] *= 2.0
    return p / np.float32(V)  # (rows need not sum to one: the decoder takes the table as is)


def _both(batch, beam, cutoff_prob, top_n, nbest, scorer=None, frame_lens=None):
    from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
    out = []
    for wave in (False, True):
        with _route(wave):
            tk, ln, sc, _ = beam_search_ids(torch.from_numpy(batch).cuda(), beam, cutoff_prob, top_n, 0, nbest=nbest,
                                            ext_scorer=scorer, frame_lens=frame_lens)
            torch.cuda.synchronize()
        out.append((tk.cpu().numpy(), ln.cpu().numpy(), sc.cpu().numpy()))
    return out


@pytest.mark.parametrize("T,V,beam,cutoff_prob,top_n,kind", [
    (150, 500, 10, 0.99, 40, "peaky"),
    (150, 4233, 10, 0.99, 40, "flat"),
    (60, 300, 16, 0.99, 64, "flat"),
    (60, 300, 1, 0.99, 40, "peaky"),
    (60, 50, 2, 1.0, 40, "flat"),      # cutoff_prob 1: every character of a small vocabulary is a candidate
    (80, 300, 5, 0.5, 3, "flat"),
    (90, 200, 10, 0.99, 40, "ties"),
    (90, 64, 16, 1.0, 64, "ties"),
    (40, 30, 7, 1.0, 40, "ties"),
])
def test_wave_kernel_equals_block_kernel(T, V, beam, cutoff_prob, top_n, kind):
    rng = np.random.Generator(np.random.PCG64(T + 3 * V + 7 * beam + top_n))
    B = 5
    if kind == "ties":
        batch = np.stack([_tie_probs(rng, T, V) for _ in range(B)])
    else:
        batch = np.stack([_probs(rng, T, V, kind) for _ in range(B)])
    lens = np.array([T, T - 1, max(T // 2, 1), 1, 0], np.int32)
    nbest = min(beam, 4)
    (t0, l0, s0), (t1, l1, s1) = _both(batch, beam, cutoff_prob, top_n, nbest, frame_lens=lens)
    assert np.array_equal(l0, l1)
    for b in range(B):
        for r in range(nbest):  # (rows of hypotheses that do not exist, lens -1, are not written by either kernel)
            if l0[b, r] >= 0:
                assert np.array_equal(t0[b, r, :l0[b, r]], t1[b, r, :l1[b, r]])
                assert s0[b, r] == s1[b, r]  # bit for bit
    if kind != "ties":  # (exact ties at the cut: upstream's order beyond (score, character) is unspecified)
        lib = _oracle()
        for b in range(B):
            ref = _oracle_decode(lib, batch[b, :lens[b]], beam, cutoff_prob, top_n, 0, 1)
            assert t1[b, 0, :l1[b, 0]].tolist() == ref[0][0]


@pytest.mark.parametrize("beam,order,alpha,beta", [(10, 3, 2.2, 4.3), (16, 2, 1.9, 0.3), (8, 5, 0.5, -1.5)])
def test_wave_kernel_equals_block_kernel_with_scorer(tmp_path, beam, order, alpha, beta):
    from ppasr_amd.decoders.beam_search_decoder import Scorer
    V, T, B = 200, 70, 4
    vocab = ["<blank>", "<unk>"] + [chr(0x4E00 + i) for i in range(V - 3)] + ["<eos>"]
    rng = np.random.Generator(np.random.PCG64(beam * 31 + order))
    known = [c for c in vocab[2:-1] if rng.random() < 0.8]
    arpa = write_synthetic_arpa(str(tmp_path / "lm.arpa"), known, order=order, seed=beam)
    scorer = Scorer(alpha, beta, arpa, vocab)
    batch = np.stack([_probs(rng, T, V, "peaky" if b % 2 else "flat") for b in range(B)])
    (t0, l0, s0), (t1, l1, s1) = _both(batch, beam, 0.99, 40, min(beam, 3), scorer=scorer)
    assert np.array_equal(l0, l1)
    ok = l0 >= 0
    assert np.array_equal(s0[ok], s1[ok])
    for b, r in zip(*np.nonzero(ok)):
        assert np.array_equal(t0[b, r, :l0[b, r]], t1[b, r, :l1[b, r]])


def test_wave_kernel_streaming_equals_one_shot_and_block_kernel():
    from ppasr_amd.decoders.beam_search_decoder import _BeamState, beam_search_ids
    rng = np.random.Generator(np.random.PCG64(77))
    T, V, beam, B = 96, 400, 10, 3
    batch = np.stack([_probs(rng, T, V, "peaky") for _ in range(B)])
    p = torch.from_numpy(batch).cuda()
    res = {}
    for wave in (False, True):
        with _route(wave):
            one = beam_search_ids(p, beam, 0.99, 40, 0, nbest=2)[:3]
            st = _BeamState(B, T, beam, p.device)
            for lo in range(0, T, 16):
                chunked = beam_search_ids(p[:, lo:lo + 16].contiguous(), beam, 0.99, 40, 0, nbest=2, state=st)[:3]
            torch.cuda.synchronize()
        L = one[0].shape[2]
        for a, b in zip(one, chunked):
            a, b = a.cpu().numpy(), b.cpu().numpy()
            if a.ndim == 3:
                b = b[:, :, :L] if b.shape[2] >= L else b
                a = a[:, :, :b.shape[2]]
            assert np.array_equal(a, b)
        res[wave] = [x.cpu().numpy() for x in one]
    for a, b in zip(res[False], res[True]):
        assert np.array_equal(a, b)
