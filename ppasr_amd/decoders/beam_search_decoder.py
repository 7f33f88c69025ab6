"""Drop-in for ``ppasr/decoders/beam_search_decoder.py`` (``BeamSearchDecoder``) and the
``paddlespeech_ctcdecoders`` wrappers of ``ppasr/decoders/swig_wrapper.py``, backed by the HIP prefix
beam search (``ppasr_ctc_beam_search`` in include/ppasr_hip.h).

External scorer: the reference always builds a KenLM ``Scorer`` (beam_search_decoder.py:28-29).  Here
``language_model_path`` may point to a character-based or word-based n-gram LM as an ARPA text file or as a KenLM binary
(``.klm``: probing / rest-probing / trie incl. the quantised and array-compressed variants; ``Scorer`` below, backed by
``ppasr_lm_*`` / ``ppasr_ctc_beam_search_lm``).  Without a model
path the search runs without a scorer (alpha / beta unused).  Returned scores follow the upstream convention:
-log P(prefix), with the LM weight removed when a scorer is used ("approx_ctc").
"""
import ctypes
import os

import numpy as np
import torch

from ppasr_amd import _lib

__all__ = ["BeamSearchDecoder", "Scorer", "ctc_beam_search_decoding", "ctc_beam_search_decoding_batch",
           "beam_search_ids"]


class Scorer:
    """swig_wrapper.py:18-33 ``Scorer(alpha, beta, model_path, vocabulary)``: external scorer for the beam search.
    The n-gram table lives on the device (hash table, csrc/lm.h); ``alpha`` / ``beta`` may be changed afterwards
    (``reset_params``) like upstream."""

    def __init__(self, alpha, beta, model_path, vocabulary, device=None):
        if not torch.cuda.is_available():
            raise _lib.PPASRHipError("no HIP device visible: ppasr_amd has no CPU fallback")
        if not os.path.exists(model_path):
            raise Exception(f"language model not found: {model_path}")
        self.alpha, self.beta = float(alpha), float(beta)
        self._lib = _lib.load()
        self._device = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        words = (ctypes.c_char_p * len(vocabulary))(*[str(w).encode("utf-8") for w in vocabulary])
        h = ctypes.c_void_p()
        with torch.cuda.device(self._device):
            _lib.check(self._lib.ppasr_lm_create(str(model_path).encode(), words, len(vocabulary), ctypes.byref(h)))
        self._h = h
        fmt = self._lib.ppasr_lm_format(h).decode()
        if fmt.startswith("klm") and not Scorer._klm_warned:
            # KenLM is not vendored with the reference and cannot be installed here: the binary layouts in csrc/klm.hip
            # are written from the KenLM sources as recalled and have only met tests/klm_writer.py, never a file produced
            # by build_binary itself.  Layout mismatches fail loudly at load; a file that loads is still unverified.
            Scorer._klm_warned = True
            import warnings
            warnings.warn(f"language model {model_path}: KenLM binary ({fmt}) read by an UNVERIFIED reader (never checked "
                          "against KenLM's own build_binary output); prefer the ARPA file of the model if you have it, and "
                          "compare a few sentence scores with kenlm.Model.score before relying on it.", RuntimeWarning,
                          stacklevel=2)

    _klm_warned = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.ppasr_lm_destroy(h)
            self._h = None

    def reset_params(self, alpha, beta):
        self.alpha, self.beta = float(alpha), float(beta)

    def get_max_order(self):
        return int(self._lib.ppasr_lm_order(self._h))

    def is_character_based(self):
        return bool(self._lib.ppasr_lm_is_character_based(self._h))

    def get_dict_size(self):
        """swig Scorer.get_dict_size(): words of a word-based model that can be spelt with the acoustic vocabulary."""
        return int(self._lib.ppasr_lm_dict_size(self._h))

    def ngram_count(self):
        return int(self._lib.ppasr_lm_ngram_count(self._h))


def _text(ids, vocabulary):
    return "".join(vocabulary[i] for i in ids)


class _BeamState:
    """Device buffer holding the beam + prefix arena of a batch of utterances (kept across chunks)."""

    def __init__(self, B, max_frames, beam_size, device):
        lib = _lib.load()
        self.bytes = int(lib.ppasr_ctc_beam_state_bytes(B, max_frames, beam_size))
        self.buf = torch.empty(self.bytes, dtype=torch.uint8, device=device)
        self.B, self.max_frames, self.beam_size = B, max_frames, beam_size
        self.frames = 0
        self.fresh = True
        self.growable = False  # streaming decoder objects: double the buffer instead of failing (the reference has no limit)

    def grow(self, need_frames):
        """Move the search into a buffer sized for at least ``need_frames`` cumulative frames (doubling)."""
        lib = _lib.load()
        cap = self.max_frames
        while cap < need_frames:
            cap *= 2
        dev = self.buf.device
        nbytes = int(lib.ppasr_ctc_beam_state_bytes(self.B, cap, self.beam_size))
        new = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.ppasr_ctc_beam_state_grow(self.buf.data_ptr(), self.bytes, new.data_ptr(), nbytes, self.B,
                                                     self.beam_size, torch.cuda.current_stream(dev).cuda_stream))
        self.buf, self.bytes, self.max_frames = new, nbytes, cap


_scratch = {}  # (device, stream) -> uint8 tensor: HBM scratch of searches whose candidate / element lists outgrow LDS


def _scratch_for(lib, dev, stream, B, T, V, beam_size, cutoff_prob, cutoff_top_n):
    """Scratch of ``ppasr_ctc_beam_scratch_bytes`` (0 bytes -> None for every configuration the reference ships).  With
    ``cutoff_prob >= 1`` -- the default of the reference's wrappers, swig_wrapper.py:38,71 -- upstream keeps every
    character of every frame; so does the kernel, through this buffer (kept and re-used per device, grown on demand)."""
    need = int(lib.ppasr_ctc_beam_scratch_bytes(B, T, V, int(beam_size), float(cutoff_prob), int(cutoff_top_n)))
    if need == 0:
        return None, 0
    key = (dev, stream)  # (searches on different streams may overlap: one buffer each)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < need:
        buf = _scratch[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    return buf, need


def beam_search_ids(probs, beam_size, cutoff_prob=1.0, cutoff_top_n=40, blank_id=0, frame_lens=None, nbest=1,
                    state=None, max_frames=None, ext_scorer=None):
    """probs [B,T,V] (numpy or device tensor) -> (tokens [B,nbest,L] i32, lens [B,nbest] i32, scores [B,nbest] f64)
    device tensors.  ``state`` (a _BeamState) continues a previous call (streaming); ``ext_scorer`` is a ``Scorer``."""
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.PPASRHipError("no HIP device visible: ppasr_amd has no CPU fallback")
    dev = probs.device if isinstance(probs, torch.Tensor) and probs.is_cuda else torch.device(
        "cuda", torch.cuda.current_device())
    p = torch.as_tensor(probs, dtype=torch.float32).to(dev).contiguous()
    B, T, V = p.shape
    if state is None:
        state = _BeamState(B, max_frames if max_frames is not None else T, beam_size, dev)
    if state.frames + T > state.max_frames:
        if not (state.growable and not state.fresh):
            raise _lib.PPASRHipError("beam-search state buffer exhausted: create the decoder with a larger max_frames")
        state.grow(state.frames + T)
    L = max(state.frames + T, 1)
    tokens = torch.empty(B, nbest, L, dtype=torch.int32, device=dev)
    lens = torch.empty(B, nbest, dtype=torch.int32, device=dev)
    scores = torch.empty(B, nbest, dtype=torch.float64, device=dev)
    fl = None if frame_lens is None else torch.as_tensor(frame_lens, dtype=torch.int32).to(dev).contiguous()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        scratch, scratch_bytes = _scratch_for(lib, dev, stream, B, T, V, beam_size, cutoff_prob, cutoff_top_n)
        _lib.check(lib.ppasr_ctc_beam_search_ws(p.data_ptr() if T > 0 else None, None if fl is None else fl.data_ptr(),
                                                B, T, V, int(beam_size), float(cutoff_prob), int(cutoff_top_n),
                                                int(blank_id), int(nbest), L, tokens.data_ptr(), lens.data_ptr(),
                                                scores.data_ptr(), state.buf.data_ptr(), state.bytes,
                                                1 if state.fresh else 0, None if ext_scorer is None else ext_scorer._h,
                                                0.0 if ext_scorer is None else ext_scorer.alpha,
                                                0.0 if ext_scorer is None else ext_scorer.beta,
                                                None if scratch is None else scratch.data_ptr(), scratch_bytes, stream))
    state.fresh = False
    state.frames += T
    return tokens, lens, scores, state


def _results(tokens, lens, scores, vocabulary, b):
    tk, ln, sc = tokens[b].cpu().numpy(), lens[b].cpu().numpy(), scores[b].cpu().numpy()
    out = []
    for r in range(tk.shape[0]):
        if ln[r] < 0:
            continue
        out.append((float(sc[r]), _text(tk[r, :ln[r]].tolist(), vocabulary)))
    return out


def ctc_beam_search_decoding(probs_seq, vocabulary, beam_size, cutoff_prob=1.0, cutoff_top_n=40, blank_id=0,
                             ext_scoring_func=None):
    """swig_wrapper.py:35-64 -> list of (score, text), best first (all ``beam_size`` hypotheses)."""
    p = probs_seq if isinstance(probs_seq, torch.Tensor) else np.asarray(probs_seq, np.float32)
    tokens, lens, scores, _ = beam_search_ids(torch.as_tensor(p)[None], beam_size, cutoff_prob, cutoff_top_n, blank_id,
                                              nbest=beam_size, ext_scorer=ext_scoring_func)
    return _results(tokens, lens, scores, vocabulary, 0)


def ctc_beam_search_decoding_batch(probs_split, vocabulary, beam_size, num_processes=1, cutoff_prob=1.0,
                                   cutoff_top_n=40, blank_id=0, ext_scoring_func=None):
    """swig_wrapper.py:67-103 -> per utterance a list of (score, text).  ``num_processes`` (CPU threads in
    the reference) is meaningless here: one workgroup per utterance."""
    if isinstance(probs_split, torch.Tensor) and probs_split.dim() == 3:
        tokens, lens, scores, _ = beam_search_ids(probs_split, beam_size, cutoff_prob, cutoff_top_n, blank_id,
                                                  nbest=beam_size, ext_scorer=ext_scoring_func)
        return [_results(tokens, lens, scores, vocabulary, b) for b in range(probs_split.shape[0])]
    shapes = {tuple(np.shape(p)) for p in probs_split}
    if len(shapes) == 1:  # equal lengths: one launch
        batch = torch.stack([torch.as_tensor(p, dtype=torch.float32) for p in probs_split])
        return ctc_beam_search_decoding_batch(batch.cuda(), vocabulary, beam_size, num_processes, cutoff_prob,
                                              cutoff_top_n, blank_id, ext_scoring_func)
    return [ctc_beam_search_decoding(p, vocabulary, beam_size, cutoff_prob, cutoff_top_n, blank_id, ext_scoring_func)
            for p in probs_split]


class BeamSearchDecoder:
    """beam_search_decoder.py:8-96 (same constructor arguments / methods)."""

    def __init__(self, alpha, beta, beam_size, cutoff_prob, cutoff_top_n, vocab_list, num_processes=10, blank_id=0,
                 language_model_path=None, max_stream_frames=5000):
        self.alpha, self.beta = alpha, beta
        self.beam_size = int(beam_size)
        self.cutoff_prob, self.cutoff_top_n = cutoff_prob, cutoff_top_n
        self.vocab_list = vocab_list
        self.num_processes = num_processes
        self.blank_id = blank_id
        # beam_search_decoder.py:19-29 downloads a default model when the path does not exist; offline that is an error
        self._ext_scorer = Scorer(alpha, beta, language_model_path, vocab_list) if language_model_path else None
        self._max_stream_frames = int(max_stream_frames)
        self._state = None

    def decode_beam_search_offline(self, probs_split):
        """-> (score, text) of the best hypothesis.  beam_search_decoder.py:45-56"""
        if self._ext_scorer is not None:
            self._ext_scorer.reset_params(self.alpha, self.beta)  # beam_search_decoder.py:46-47
        p = probs_split if isinstance(probs_split, torch.Tensor) else np.asarray(probs_split, np.float32)
        tokens, lens, scores, _ = beam_search_ids(torch.as_tensor(p)[None], self.beam_size, self.cutoff_prob,
                                                  self.cutoff_top_n, self.blank_id, nbest=1, ext_scorer=self._ext_scorer)
        return _results(tokens, lens, scores, self.vocab_list, 0)[0]

    def decode_batch_beam_search_offline(self, probs_split):
        """-> list[str].  beam_search_decoder.py:59-73 (every row of every table is decoded)."""
        if self._ext_scorer is not None:
            self._ext_scorer.reset_params(self.alpha, self.beta)  # beam_search_decoder.py:60-61
        if isinstance(probs_split, torch.Tensor) and probs_split.dim() == 3:
            # only the best hypothesis is used (beam_search_decoder.py:72): ask the kernel for nbest = 1 instead of
            # copying and stringifying all beam_size hypotheses of every utterance
            tokens, lens, scores, _ = beam_search_ids(probs_split, self.beam_size, self.cutoff_prob, self.cutoff_top_n,
                                                      self.blank_id, nbest=1, ext_scorer=self._ext_scorer)
            tk, ln = tokens[:, 0].cpu(), lens[:, 0].cpu()
            return [_text(tk[b, :max(int(ln[b]), 0)].tolist(), self.vocab_list) for b in range(tk.shape[0])]
        res = ctc_beam_search_decoding_batch(probs_split, self.vocab_list, self.beam_size, self.num_processes,
                                             self.cutoff_prob, self.cutoff_top_n, self.blank_id, self._ext_scorer)
        return [r[0][1] for r in res]

    def decode_chunk(self, probs, logits_lens):
        """Streaming: feed one chunk [B=1,c,V], return the current best (score, text).
        beam_search_decoder.py:75-91"""
        p = torch.as_tensor(probs, dtype=torch.float32)
        if p.dim() == 2:
            p = p[None]
        if self._state is None:
            dev = p.device if p.is_cuda else torch.device("cuda", torch.cuda.current_device())
            self._state = _BeamState(p.shape[0], max(self._max_stream_frames, p.shape[1]), self.beam_size, dev)
            self._state.growable = True
        lens = np.asarray(logits_lens).astype(np.int32)
        tokens, ln, scores, _ = beam_search_ids(p, self.beam_size, self.cutoff_prob, self.cutoff_top_n, self.blank_id,
                                                frame_lens=lens, nbest=1, state=self._state, ext_scorer=self._ext_scorer)
        return _results(tokens, ln, scores, self.vocab_list, 0)[0]

    def reset_decoder(self):
        """beam_search_decoder.py:93-96"""
        self._state = None
