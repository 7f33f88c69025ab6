"""GPU parity of the HIP prefix beam search against the C oracle (oracle/ctc_beam_search_oracle.c,
a restatement of the upstream paddlespeech_ctcdecoders algorithm -- parity UNPINNED by the
reference, see the oracle header).  Token sequences must match exactly; scores to float round-off
(logf/expf differ by <= 1 ulp between glibc and the GPU math library)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libctc_beam_oracle.so"))
    lib.ctc_beam_oracle_decode.restype = ctypes.c_int
    lib.ctc_beam_oracle_create.restype = ctypes.c_void_p
    lib.ctc_beam_oracle_next.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.ctc_beam_oracle_result.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p]
    lib.ctc_beam_oracle_free.argtypes = [ctypes.c_void_p]
    return lib


def _oracle_decode(lib, p, beam, cutoff_prob, top_n, blank, nbest):
    T, V = p.shape
    L = max(T, 1)
    tokens = np.empty((nbest, L), np.int32)
    lens = np.empty(nbest, np.int32)
    scores = np.empty(nbest, np.float64)
    p = np.ascontiguousarray(p, np.float32)
    n = lib.ctc_beam_oracle_decode(p.ctypes.data_as(ctypes.c_void_p), T, V, beam, ctypes.c_double(cutoff_prob), top_n,
                                   blank, nbest, L, tokens.ctypes.data_as(ctypes.c_void_p),
                                   lens.ctypes.data_as(ctypes.c_void_p), scores.ctypes.data_as(ctypes.c_void_p))
    return [(tokens[i, :lens[i]].tolist(), scores[i]) for i in range(n)]


def _probs(rng, T, V, kind):
    logits = rng.standard_normal((T, V)).astype(np.float32)
    if kind == "peaky":
        idx = np.repeat(rng.integers(0, V, size=(T + 2) // 3), 3)[:T]
        logits[np.arange(T), idx] += 6.0
        logits[:, 0] += np.where(rng.random(T) < 0.4, 7.0, 0.0).astype(np.float32)
    elif kind == "flat":
        logits *= 0.3
    e = np.exp(logits - logits.max(1, keepdims=True))
    return (e / e.sum(1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("T,V,beam,cutoff_prob,top_n", [
    (50, 30, 8, 1.0, 40),        # english-style config: no pruning (cutoff_prob 1.0, V < top_n)
    (120, 500, 10, 0.99, 40),    # configs[3]-style: beam 10
    (249, 4233, 10, 0.99, 40),
    (60, 4233, 64, 0.99, 40),
    (40, 200, 300, 0.99, 40),    # the reference's default beam_size (conformer.yml:82)
    (1, 100, 5, 0.99, 40),
])
@pytest.mark.parametrize("kind", ["peaky", "flat"])
def test_beam_search_matches_c_oracle(T, V, beam, cutoff_prob, top_n, kind):
    from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
    lib = _oracle()
    rng = np.random.Generator(np.random.PCG64(T * 7 + V + beam))
    B = 3
    batch = np.stack([_probs(rng, T, V, kind) for _ in range(B)])
    nbest = min(beam, 5)
    tokens, lens, scores, _ = beam_search_ids(torch.from_numpy(batch).cuda(), beam, cutoff_prob, top_n, 0, nbest=nbest)
    torch.cuda.synchronize()
    tokens, lens, scores = tokens.cpu().numpy(), lens.cpu().numpy(), scores.cpu().numpy()
    for b in range(B):
        ref = _oracle_decode(lib, batch[b], beam, cutoff_prob, top_n, 0, nbest)
        assert tokens[b, 0, :lens[b, 0]].tolist() == ref[0][0], (b, ref[0])
        assert abs(scores[b, 0] - ref[0][1]) <= 1e-4 * max(1.0, abs(ref[0][1]))
        # the rest of the n-best list: same hypotheses unless two scores are within float noise
        for r in range(1, len(ref)):
            if abs(ref[r][1] - ref[r - 1][1]) > 1e-4 and (r + 1 >= len(ref) or abs(ref[r + 1][1] - ref[r][1]) > 1e-4):
                assert tokens[b, r, :lens[b, r]].tolist() == ref[r][0], (b, r)


def test_streaming_chunks_equal_offline_and_oracle_object():
    """CtcBeamSearchDecoderBatch semantics: next(chunk) ... decode() == one-shot decode of the whole table."""
    from ppasr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    lib = _oracle()
    rng = np.random.Generator(np.random.PCG64(5))
    V, beam = 300, 10
    vocab = ["<blank>"] + [chr(0x4E00 + i) for i in range(V - 1)]
    p = _probs(rng, 100, V, "peaky")
    dec = BeamSearchDecoder(2.2, 4.3, beam, 0.99, 40, vocab)
    off_score, off_text = dec.decode_beam_search_offline(p)
    h = lib.ctc_beam_oracle_create(V, beam, ctypes.c_double(0.99), 40, 0)
    text = None
    for s in range(0, 100, 16):
        chunk = np.ascontiguousarray(p[s:s + 16])
        score, text = dec.decode_chunk(chunk[None], np.array([chunk.shape[0]]))
        lib.ctc_beam_oracle_next(h, chunk.ctypes.data_as(ctypes.c_void_p), chunk.shape[0])
        tk = np.empty((1, 200), np.int32); ln = np.empty(1, np.int32); sc = np.empty(1, np.float64)
        lib.ctc_beam_oracle_result(h, 1, 200, tk.ctypes.data_as(ctypes.c_void_p), ln.ctypes.data_as(ctypes.c_void_p),
                                   sc.ctypes.data_as(ctypes.c_void_p))
        assert text == "".join(vocab[i] for i in tk[0, :ln[0]])
    lib.ctc_beam_oracle_free(h)
    assert text == off_text and abs(score - off_score) < 1e-4 * max(1.0, abs(off_score))
    dec.reset_decoder()
    assert dec.decode_chunk(p[None, :16], np.array([16]))[1] != ""  # fresh state after reset


def test_streaming_past_the_sized_capacity_grows_the_state():
    """The reference's decoder object has no frame limit: a stream longer than `max_stream_frames` moves the beams and
    prefix arenas into a doubled buffer (ppasr_ctc_beam_state_grow) and continues the SAME search -- every chunk's best
    prefix equals the one of an object sized for the whole stream, the final one the offline decode."""
    from ppasr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    rng = np.random.Generator(np.random.PCG64(15))
    V, beam = 200, 8
    vocab = ["<blank>"] + [chr(0x4E00 + i) for i in range(V - 1)]
    p = _probs(rng, 150, V, "peaky")
    small = BeamSearchDecoder(2.2, 4.3, beam, 0.99, 40, vocab, max_stream_frames=16)
    big = BeamSearchDecoder(2.2, 4.3, beam, 0.99, 40, vocab, max_stream_frames=400)
    off_score, off_text = big.decode_beam_search_offline(p)
    for s in range(0, 150, 13):
        chunk = np.ascontiguousarray(p[s:s + 13])
        a = small.decode_chunk(chunk[None], np.array([chunk.shape[0]]))
        b = big.decode_chunk(chunk[None], np.array([chunk.shape[0]]))
        assert a[1] == b[1] and abs(a[0] - b[0]) < 1e-9 * max(1.0, abs(b[0])), s
    assert small._state.max_frames >= 150 and a[1] == off_text


def test_frame_lens_and_batch_api():
    from ppasr_amd.decoders.beam_search_decoder import BeamSearchDecoder, beam_search_ids
    lib = _oracle()
    rng = np.random.Generator(np.random.PCG64(9))
    V, beam, T = 120, 6, 40
    vocab = ["<blank>"] + [chr(0x4E00 + i) for i in range(V - 1)]
    batch = np.stack([_probs(rng, T, V, "peaky") for _ in range(4)])
    lens = np.array([40, 17, 0, 5], np.int32)
    tokens, ln, sc, _ = beam_search_ids(torch.from_numpy(batch).cuda(), beam, 0.99, 40, 0, frame_lens=lens)
    for b in range(4):
        ref = _oracle_decode(lib, batch[b, :lens[b]], beam, 0.99, 40, 0, 1)
        assert tokens[b, 0, :int(ln[b, 0])].tolist() == ref[0][0]
    dec = BeamSearchDecoder(2.2, 4.3, beam, 0.99, 40, vocab)
    texts = dec.decode_batch_beam_search_offline(list(batch))
    for b in range(4):
        ref = _oracle_decode(lib, batch[b], beam, 0.99, 40, 0, 1)
        assert texts[b] == "".join(vocab[i] for i in ref[0][0])


def test_unpruned_configuration_keeps_every_character():
    """cutoff_prob = 1.0 (the default of the Python wrappers, swig_wrapper.py:38,71): upstream then ignores cutoff_top_n and
    keeps all V characters of every frame -- so does the kernel (wide pruning records and, where beam x (1 + V) entries do
    not fit LDS, the element list in HBM scratch).  Equal to the oracle's genuinely unpruned decode; the plain C entry point
    without scratch refuses the call instead of truncating it."""
    import ctypes as C
    from ppasr_amd import _lib
    from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
    lib = _oracle()
    hl = _lib.load()
    assert not hasattr(hl, "ppasr_ctc_beam_candidate_cap")
    rng = np.random.Generator(np.random.PCG64(4242))
    T, V, beam = 40, 4233, 10
    assert hl.ppasr_ctc_beam_scratch_bytes(2, T, V, beam, 0.99, 40) == 0
    assert hl.ppasr_ctc_beam_scratch_bytes(2, T, V, beam, 1.0, 40) >= 2 * T * (2 + 2 * V) * 4
    batch = np.stack([_probs(rng, T, V, kind) for kind in ("peaky", "flat")])
    dev = torch.from_numpy(batch).cuda()
    tokens, lens, scores, _ = beam_search_ids(dev, beam, 1.0, 40, 0, nbest=1)
    torch.cuda.synchronize()
    for b in range(2):
        got = tokens[b, 0, :int(lens[b, 0])].cpu().tolist()
        unpruned = _oracle_decode(lib, batch[b], beam, 1.0, 40, 0, 1)
        assert got == unpruned[0][0]
        assert abs(float(scores[b, 0]) - unpruned[0][1]) < 1e-4 * max(1.0, abs(unpruned[0][1]))
    # the entry points without scratch: PPASR_ENOSPACE, nothing silently dropped
    nbytes = int(hl.ppasr_ctc_beam_state_bytes(2, T, beam))
    st = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    tk = torch.empty(2, 1, T, dtype=torch.int32, device="cuda")
    ln = torch.empty(2, 1, dtype=torch.int32, device="cuda")
    sc = torch.empty(2, 1, dtype=torch.float64, device="cuda")
    rc = hl.ppasr_ctc_beam_search(dev.data_ptr(), None, 2, T, V, beam, C.c_double(1.0), 40, 0, 1, T, tk.data_ptr(),
                                  ln.data_ptr(), sc.data_ptr(), st.data_ptr(), nbytes, 1, None)
    assert rc == _lib.PPASR_ENOSPACE


class _beam_fast:
    """PPASR_BEAM_FAST is read at every call: 1 (default) = small beams rank the staircase-restricted element list first
    (ctc_beam.hip (e')), 0 = always the general selection."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("PPASR_BEAM_FAST")
        os.environ["PPASR_BEAM_FAST"] = "1" if self.on else "0"

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("PPASR_BEAM_FAST", None)
        else:
            os.environ["PPASR_BEAM_FAST"] = self.old


def _tied_probs(rng, T, V):
    """Tables with EXACT ties: probabilities drawn from a handful of values (equal candidates inside a frame, equal scores
    between hypotheses), frames that repeat, frames where the blank takes nearly everything."""
    vals = np.array([1.0, 2.0, 2.0, 4.0, 8.0, 8.0, 16.0], np.float32)
    w = vals[rng.integers(0, len(vals), size=(T, V))]
    w[:, 0] *= np.where(rng.random(T) < 0.3, 400.0, 1.0).astype(np.float32)
    for t in range(1, T):
        if rng.random() < 0.25:
            w[t] = w[t - 1]
    return (w / w.sum(1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("T,V,beam,cutoff_prob,top_n", [
    (249, 4233, 10, 0.99, 40),   # BASELINE configs[3] / [4]
    (200, 500, 10, 0.99, 40),
    (150, 90, 16, 0.999, 40),
    (120, 64, 5, 0.99, 12),
    (100, 300, 1, 0.99, 40),
    (80, 50, 8, 1.0, 40),        # no pruning: every character of every frame
    (60, 700, 13, 0.9, 7),
])
@pytest.mark.parametrize("kind", ["peaky", "flat", "tied"])
def test_staircase_fast_path_is_bit_identical_to_the_general_selection(T, V, beam, cutoff_prob, top_n, kind):
    """Same survivors, same order, same node ids: every n-best token sequence and every score bit for bit, with and without
    the fast path -- and the best path equal to the C oracle's."""
    from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
    lib = _oracle()
    rng = np.random.Generator(np.random.PCG64(T * 11 + V + beam))
    B = 4
    batch = np.stack([_tied_probs(rng, T, V) if kind == "tied" else _probs(rng, T, V, kind) for _ in range(B)])
    lens = np.array([T, T - 7, T // 2, 1], np.int32)
    nbest = beam
    outs = []
    for on in (True, False):
        with _beam_fast(on):
            tokens, ln, sc, _ = beam_search_ids(torch.from_numpy(batch).cuda(), beam, cutoff_prob, top_n, 0, nbest=nbest,
                                                frame_lens=lens)
            torch.cuda.synchronize()
        outs.append((tokens.cpu().numpy(), ln.cpu().numpy(), sc.cpu().numpy()))
    assert np.array_equal(outs[0][1], outs[1][1])
    for b in range(B):
        for r in range(nbest):
            n = outs[0][1][b, r]
            if n < 0:  # fewer hypotheses than nbest (short utterances): the row is not written
                continue
            assert np.array_equal(outs[0][0][b, r, :n], outs[1][0][b, r, :n]), (b, r)
            assert outs[0][2][b, r] == outs[1][2][b, r], (b, r)  # float64 scores: the same sums in the same order
    tokens, ln, sc = outs[0]
    if kind == "tied":  # (exact score ties at the cut: the oracle's container order decides there; only the two selections
        return          #  of the kernel are compared on these tables)
    for b in range(B):
        ref = _oracle_decode(lib, batch[b, :lens[b]], beam, cutoff_prob, top_n, 0, 1)
        assert tokens[b, 0, :ln[b, 0]].tolist() == ref[0][0], b
        assert abs(sc[b, 0] - ref[0][1]) <= 1e-4 * max(1.0, abs(ref[0][1]))


def test_staircase_fast_path_streaming_state_is_identical():
    """Chunked decoding (state carried between calls): the persisted beams of both selections agree after every chunk."""
    from ppasr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    rng = np.random.Generator(np.random.PCG64(77))
    V, beam = 300, 10
    vocab = ["<blank>"] + [chr(0x4E00 + i) for i in range(V - 1)]
    p = _probs(rng, 160, V, "peaky")
    res = []
    for on in (True, False):
        with _beam_fast(on):
            dec = BeamSearchDecoder(2.2, 4.3, beam, 0.99, 40, vocab)
            out = []
            for s0 in range(0, 160, 16):
                out.append(dec.decode_chunk(p[None, s0:s0 + 16], np.array([16])))
        res.append(out)
    assert res[0] == res[1]


@pytest.mark.parametrize("beam,V,T", [(10, 300, 120), (16, 90, 100), (100, 500, 80), (300, 4233, 60)])
@pytest.mark.parametrize("kind", ["peaky", "tied", "flat"])
def test_clipped_rows_without_margin_fall_back_to_full_rows(beam, V, T, kind):
    """PPASR_BEAM_MARGIN=0 (rows clipped to the bare (rank + 1)(k + 1) <= beam staircase): the verification now fails on a
    share of the frames, so the fall-back -- full rows after a clipped attempt, including the tiny-beam path's merged last
    phase being undone -- is exercised; results must still equal the full-row selection bit for bit."""
    from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
    rng = np.random.Generator(np.random.PCG64(beam * 3 + V))
    B = 3
    batch = np.stack([_tied_probs(rng, T, V) if kind == "tied" else _probs(rng, T, V, kind) for _ in range(B)])
    dev = torch.from_numpy(batch).cuda()
    old = os.environ.get("PPASR_BEAM_MARGIN")
    os.environ["PPASR_BEAM_MARGIN"] = "0"
    try:
        with _beam_fast(True):
            a = beam_search_ids(dev, beam, 0.99, 40, 0, nbest=min(beam, 8))
            torch.cuda.synchronize()
    finally:
        if old is None:
            os.environ.pop("PPASR_BEAM_MARGIN", None)
        else:
            os.environ["PPASR_BEAM_MARGIN"] = old
    with _beam_fast(False):
        b = beam_search_ids(dev, beam, 0.99, 40, 0, nbest=min(beam, 8))
        torch.cuda.synchronize()
    assert torch.equal(a[1], b[1])
    ln = a[1].cpu().numpy()
    ta, tb = a[0].cpu().numpy(), b[0].cpu().numpy()
    for u in range(B):
        for r in range(ln.shape[1]):
            n = ln[u, r]
            if n >= 0:
                assert np.array_equal(ta[u, r, :n], tb[u, r, :n]), (u, r)
    assert torch.equal(a[2], b[2])
