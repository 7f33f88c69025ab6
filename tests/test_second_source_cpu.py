"""CPU: second sources for the rows that cannot be pinned to the reference (their arithmetic lives in third-party modules
that are not in /root/reference).  They do NOT replace a pin; they check each restatement against a formulation that was
not written from the same reading:

(a) beam search -- the C oracle (oracle/ctc_beam_search_oracle.c: upstream's prefix TRIE with node bookkeeping, float32)
    against a dictionary-of-prefixes formulation ({prefix tuple: (log P_b, log P_nb)} per frame, no trie, no node
    identity, float64) on random tables, with pruning disabled and with top-n / cumulative pruning.  Proves: the trie
    bookkeeping (exists flags, revival of removed nodes, prev / cur swapping, repeated-character rule) computes the
    prefix probabilities of the textbook recursion.  Does not prove: that upstream's pruning rule or score convention is
    what the oracle says (both sides take the rule from the same recollection).
(b) fbank -- oracle/fbank_oracle.py (numpy float64, explicit frame matrix + np.fft) against an independent construction
    on torch: strided framing, the pre-emphasis / DC removal folded into one banded operator, torch.fft, a mel filter
    bank built from bin EDGES in Hz mapped through the inverse mel scale.  Proves: framing, window, FFT size, filter
    shapes agree between two derivations of Kaldi's published algorithm.  Does not prove: paddleaudio's defaults.
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLT_MIN = float(np.finfo(np.float32).tiny)


def _oracle_lib():
    so = os.path.join(ROOT, "oracle", "_build", "libctc_beam_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    lib.ctc_beam_oracle_decode.restype = ctypes.c_int
    return lib


def _oracle_top(lib, p, beam, cutoff_prob, top_n, nbest=1):
    T, V = p.shape
    L = max(T, 1)
    tokens = np.empty((nbest, L), np.int32)
    lens = np.empty(nbest, np.int32)
    scores = np.empty(nbest, np.float64)
    p = np.ascontiguousarray(p, np.float32)
    n = lib.ctc_beam_oracle_decode(p.ctypes.data_as(ctypes.c_void_p), T, V, beam, ctypes.c_double(cutoff_prob), top_n, 0,
                                   nbest, L, tokens.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p),
                                   scores.ctypes.data_as(ctypes.c_void_p))
    return [(tuple(tokens[i, :lens[i]].tolist()), float(scores[i])) for i in range(n)]


def _lse(a, b):
    if a == -math.inf:
        return b
    if b == -math.inf:
        return a
    m = max(a, b)
    return m + math.log(math.exp(a - m) + math.exp(b - m))


def _prefix_dict_search(p, beam, cutoff_prob, top_n, blank=0):
    """CTC prefix beam search over a dict {prefix: [log P_blank, log P_nonblank]} (float64): every frame, every kept
    prefix spreads its mass to itself (blank, repeated last character) and to its one-character extensions (from P_b
    only when the character repeats the last one); the best `beam` prefixes by log(P_b + P_nb) are kept."""
    T, V = p.shape
    beam_set = {(): [0.0, -math.inf]}
    for t in range(T):
        row = p[t].astype(np.float64)
        order = sorted(range(V), key=lambda i: (-row[i], i))
        if cutoff_prob < 1.0:
            cum, keep = 0.0, []
            for i in order:
                cum += row[i]
                keep.append(i)
                if cum >= cutoff_prob or len(keep) >= top_n:
                    break
        else:
            keep = order
        logp = {c: math.log(row[c] + FLT_MIN) for c in keep}
        nxt = {}
        for pre, (pb, pnb) in beam_set.items():
            tot = _lse(pb, pnb)
            for c in keep:
                lp = logp[c]
                if c == blank:
                    e = nxt.setdefault(pre, [-math.inf, -math.inf])
                    e[0] = _lse(e[0], lp + tot)
                    continue
                if pre and c == pre[-1]:
                    e = nxt.setdefault(pre, [-math.inf, -math.inf])
                    e[1] = _lse(e[1], lp + pnb)
                    if pb > -math.inf:
                        e2 = nxt.setdefault(pre + (c,), [-math.inf, -math.inf])
                        e2[1] = _lse(e2[1], lp + pb)
                else:
                    e2 = nxt.setdefault(pre + (c,), [-math.inf, -math.inf])
                    e2[1] = _lse(e2[1], lp + tot)
        # prefixes of the previous beam that received nothing stay in upstream's trie with probability 0: irrelevant
        ranked = sorted(nxt.items(), key=lambda kv: (-_lse(*kv[1]), kv[0][-1] if kv[0] else -1))
        beam_set = dict(ranked[:beam])
    ranked = sorted(beam_set.items(), key=lambda kv: (-_lse(*kv[1]), kv[0][-1] if kv[0] else -1))
    return [(k, -_lse(*v)) for k, v in ranked]


@pytest.mark.parametrize("cutoff_prob,top_n", [(1.0, 40), (0.99, 40), (0.9, 5)])
def test_beam_oracle_equals_prefix_dictionary_formulation(cutoff_prob, top_n):
    lib = _oracle_lib()
    rng = np.random.Generator(np.random.PCG64(int(cutoff_prob * 100) + top_n))
    same = checked = 0
    for case in range(200):
        T, V, beam = int(rng.integers(1, 25)), int(rng.integers(3, 40)), int(rng.choice([1, 2, 5, 10, 30]))
        sharp = float(rng.choice([0.5, 2.0, 5.0]))
        logits = rng.standard_normal((T, V)) * sharp
        p = np.exp(logits - logits.max(-1, keepdims=True))
        p = (p / p.sum(-1, keepdims=True)).astype(np.float32)
        ref = _oracle_top(lib, p, beam, cutoff_prob, top_n, nbest=min(beam, 3))
        got = _prefix_dict_search(p, beam, cutoff_prob, top_n)
        checked += 1
        if got[0][0] == ref[0][0]:
            same += 1
            assert abs(got[0][1] - ref[0][1]) <= 2e-4 * max(1.0, abs(ref[0][1])), (case, got[0], ref[0])
        else:
            # float32 trie vs float64 dictionary: a different winner is only acceptable on a near-tie of the two leaders
            alt = dict(got).get(ref[0][0])
            assert alt is not None and abs(alt - got[0][1]) <= 1e-3 * max(1.0, abs(alt)), (case, got[:2], ref[:2])
    assert same >= checked - 3, (same, checked)


# ---------------------------------------------------------------------------------------------------------------------
def _independent_fbank(int16_samples, sr=16000, n_mels=80, frame_ms=25.0, shift_ms=10.0):
    x = torch.as_tensor(np.asarray(int16_samples), dtype=torch.float64)
    win, hop = int(round(sr * frame_ms / 1000.0)), int(round(sr * shift_ms / 1000.0))
    if x.numel() < win:
        return torch.zeros(0, n_mels, dtype=torch.float64)
    frames = x.unfold(0, win, hop)                                  # snip_edges: only whole frames
    nfft = 1
    while nfft < win:
        nfft *= 2
    # DC removal and pre-emphasis as ONE linear operator on a frame: y = P (I - 11^T / win) f, P = I - 0.97 * shift
    # (with the first sample's predecessor replicated, Kaldi's convention)
    eye = torch.eye(win, dtype=torch.float64)
    center = eye - torch.full((win, win), 1.0 / win, dtype=torch.float64)
    pre = eye.clone()
    pre[torch.arange(1, win), torch.arange(0, win - 1)] -= 0.97
    pre[0, 0] -= 0.97
    n = torch.arange(win, dtype=torch.float64)
    povey = torch.pow(0.5 * (1.0 - torch.cos(2.0 * math.pi * n / (win - 1))), 0.85)
    op = torch.diag(povey) @ pre @ center
    spec = torch.fft.rfft(frames @ op.T, n=nfft, dim=1)
    power = spec.real ** 2 + spec.imag ** 2
    # mel filters from band EDGES in Hz: equally spaced on the mel axis between 20 Hz and Nyquist, triangular in mel
    inv_mel = lambda m: 700.0 * (math.exp(m / 1127.0) - 1.0)
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    lo, hi = mel(20.0), mel(sr / 2.0)
    edges_hz = [inv_mel(lo + (hi - lo) * k / (n_mels + 1)) for k in range(n_mels + 2)]
    fb = torch.zeros(n_mels, nfft // 2, dtype=torch.float64)
    for b in range(n_mels):
        ml, mc, mr = mel(edges_hz[b]), mel(edges_hz[b + 1]), mel(edges_hz[b + 2])
        for k in range(nfft // 2):
            m = mel(k * sr / nfft)
            if ml < m < mr:
                fb[b, k] = (m - ml) / (mc - ml) if m <= mc else (mr - m) / (mr - mc)
    e = power[:, : nfft // 2] @ fb.T
    return torch.log(torch.clamp(e, min=float(np.finfo(np.float32).eps)))


@pytest.mark.parametrize("seconds,sr", [(1.0, 16000), (0.31, 16000), (0.024, 16000), (0.5, 8000)])
def test_fbank_oracle_equals_independent_construction(seconds, sr):
    from oracle.fbank_oracle import kaldi_fbank
    rng = np.random.Generator(np.random.PCG64(int(seconds * 1000) + sr))
    t = np.arange(int(seconds * sr)) / sr
    wav = 0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 3100 * t + 1.0) + 0.05 * rng.standard_normal(t.shape)
    pcm = np.clip(wav * 32768.0, -32768, 32767).astype(np.int16)
    a = kaldi_fbank(pcm, sr=sr)
    b = _independent_fbank(pcm, sr=sr).numpy()
    assert a.shape == b.shape
    if a.size:
        assert a.shape[1] == 80 and a.shape[0] == 1 + (len(pcm) - int(sr * 0.025)) // int(sr * 0.010)
        assert float(np.abs(a - b).max()) < 1e-5, float(np.abs(a - b).max())


# --------------------------------------------------------------------------------------------------------------------
# (c) recurrent cells -- oracle/paddle_shim's nn.LSTM / nn.GRU (the arithmetic the DeepSpeech2 reference-source fixtures
#     run on: gate order, `h = z h + (1 - z) c`, `sequence_length` freezing; written from Paddle's documentation) against
#     torch.nn.LSTM / torch.nn.GRU with the same weights, bidirectional, padded batches as packed sequences.  torch's
#     cells are an independent implementation (ATen C++, fused gate kernels) of the same documented equations.
#     Proves: the shim's step arithmetic, the reverse direction starting at len - 1, zeros behind len and the final states.
#     Does not prove: that Paddle's gate order / candidate formula equals torch's (both are the cuDNN convention as far as
#     either project documents it).
def _shim_paddle():
    import sys
    shim = os.path.join(ROOT, "oracle", "paddle_shim")
    if shim not in sys.path:
        sys.path.insert(0, shim)
    import paddle  # noqa: F401  (the torch-backed shim)
    return paddle


@pytest.mark.parametrize("kind", ["LSTM", "GRU"])
@pytest.mark.parametrize("direction", ["forward", "bidirect"])
@pytest.mark.parametrize("ragged", [False, True])
def test_shim_rnn_equals_torch_rnn(kind, direction, ragged):
    paddle = _shim_paddle()
    torch.manual_seed(11)
    B, T, I, H = 3, 17, 24, 32
    shim = getattr(paddle.nn, kind)(I, H, direction=direction)
    bi = direction != "forward"
    ref = getattr(torch.nn, kind)(I, H, num_layers=1, batch_first=True, bidirectional=bi).double()
    with torch.no_grad():
        for sfx in ([""] + (["_reverse"] if bi else [])):
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                getattr(ref, f"{n}_l0{sfx}").copy_(getattr(shim, f"{n}_l0{sfx}")._t.double())
    x = torch.randn(B, T, I)
    lens = torch.tensor([T, 9, 1]) if ragged else torch.tensor([T] * B)
    nd = 2 if bi else 1
    h0 = torch.randn(nd, B, H) * 0.5
    c0 = torch.randn(nd, B, H) * 0.5
    init = (paddle.to_tensor(h0.numpy()), paddle.to_tensor(c0.numpy())) if kind == "LSTM" else paddle.to_tensor(h0.numpy())
    y, st = shim(paddle.to_tensor(x.numpy()), init, paddle.to_tensor(lens.numpy()) if ragged else None)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x.double(), lens, batch_first=True, enforce_sorted=True)
    with torch.no_grad():
        yp, st_ref = ref(packed, (h0.double(), c0.double()) if kind == "LSTM" else h0.double())
    y_ref, _ = torch.nn.utils.rnn.pad_packed_sequence(yp, batch_first=True, total_length=T)
    assert (y._t.double() - y_ref).abs().max().item() < 1e-6
    for b in range(B):  # zeros behind len, both directions
        assert y._t[b, int(lens[b]):].abs().max().item() == 0 if int(lens[b]) < T else True
    finals = st if kind == "LSTM" else (st,)
    finals_ref = st_ref if kind == "LSTM" else (st_ref,)
    for a, b_ in zip(finals, finals_ref):
        assert (a._t.double() - b_).abs().max().item() < 1e-6


def test_shim_rnn_state_dict_names_are_paddles():
    """RNNBase registers each tensor as weight_ih_l0[_reverse] AND under 0.cell[_fw|_bw] (the names the reference's
    checkpoints carry, deepspeech2/encoder.py:36-48)."""
    paddle = _shim_paddle()
    names = set(paddle.nn.LSTM(4, 8, direction="bidirect").state_dict().keys())
    for n in ("weight_ih_l0", "weight_hh_l0_reverse", "0.cell_fw.weight_ih", "0.cell_bw.bias_hh"):
        assert n in names, (n, sorted(names))


# ---- the mean square of AudioSegment.rms_db: numpy's summation order, restated element by element ----
def _pairwise_f32(a):
    """numpy's float32 pairwise sum (what csrc/fbank.hip k_sumsq reproduces): <= 128 elements -> eight running sums combined
    ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n % 8 tail; longer -> split at n/2 rounded down to a multiple of 8."""
    f32 = np.float32
    n = len(a)
    if n < 8:
        r = f32(0.0)
        for v in a:
            r = f32(r + v)
        return r
    if n <= 128:
        r = [a[j] for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = f32(r[j] + a[i + j])
            i += 8
        res = f32(f32(f32(r[0] + r[1]) + f32(r[2] + r[3])) + f32(f32(r[4] + r[5]) + f32(r[6] + r[7])))
        while i < n:
            res = f32(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f32(_pairwise_f32(a[:n2]) + _pairwise_f32(a[n2:]))


def test_numpy_sums_float32_in_8192_element_chunks_of_pairwise_sums():
    """The GPU fbank front end computes AudioSegment.normalize's gain bit-exactly because it sums the squares in numpy's
    order: `np.mean(x ** 2)` on float32 = sequential accumulation of pairwise sums over 8192-element chunks (the ufunc
    buffer size), divided once.  If a numpy release changes that order this test says so before the GPU test does."""
    import warnings
    from oracle.fbank_oracle import normalize_to_int16
    from ppasr_amd.data_utils.featurizer import db_gain
    f32 = np.float32
    for n in (1, 7, 8, 129, 1000, 8192, 8193, 20000, 40000):
        x = np.random.default_rng(n).standard_normal(n).astype(f32) * f32(0.1)
        a = x ** 2
        s = f32(0.0)
        for i in range(0, n, 8192):
            s = f32(s + _pairwise_f32(a[i:i + 8192]))
        assert s == np.sum(a), n
        ms = f32(np.float64(s) / n)
        assert ms == np.mean(a), n
        # ... and the gain on top of it with numpy 1.x's scalar types (float32 mean square and log10, everything after in float64)
        rms_db = 10.0 * float(np.log10(ms))
        gain = f32(10.0 ** ((-20.0 - rms_db) / 20.0))
        assert gain == db_gain(x, -20.0), n
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = np.clip((x * gain) * f32(32768.0), -32768, 32767).astype(np.int16)
        assert np.array_equal(normalize_to_int16(x, True, -20.0), want), n
