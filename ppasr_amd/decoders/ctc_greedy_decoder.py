"""Drop-in for ``ppasr/decoders/ctc_greedy_decoder.py`` backed by the HIP kernels
(``ppasr_ctc_greedy`` in include/ppasr_hip.h): same names, argument meaning and return values.

``probs_seq`` may be a numpy array (copied to the GPU) or a device tensor (used in place,
which is the point: the reference copies the whole [T', V] table to the host first).
Token ids -> text happens on the host with the caller's vocabulary list.
"""
import numpy as np
import torch

from ppasr_amd import _lib

__all__ = ["greedy_decoder", "greedy_decoder_batch", "greedy_decoder_chunk", "greedy_decode_ids"]


def _device():
    if not torch.cuda.is_available():
        raise _lib.PPASRHipError("no HIP device visible: ppasr_amd has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def greedy_decode_ids(probs, frame_lens=None, blank_index=0):
    """probs [B, T, V] -> (tokens [B,T] i32 (-1 padded), n_tokens [B] i32, score [B] f64, frame_argmax
    [B,T] i32, frame_maxprob [B,T] f32), all device tensors."""
    lib = _lib.load()
    dev = probs.device if isinstance(probs, torch.Tensor) and probs.is_cuda else _device()
    p = torch.as_tensor(probs, dtype=torch.float32).to(dev).contiguous()
    B, T, V = p.shape
    tokens = torch.empty(B, T, dtype=torch.int32, device=dev)
    n_tokens = torch.empty(B, dtype=torch.int32, device=dev)
    score = torch.empty(B, dtype=torch.float64, device=dev)
    ws = torch.empty(B * T * 2, dtype=torch.int32, device=dev)
    fl = None if frame_lens is None else torch.as_tensor(frame_lens, dtype=torch.int32).to(dev).contiguous()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.ppasr_ctc_greedy(p.data_ptr(), None if fl is None else fl.data_ptr(), B, T, V, blank_index,
                                        tokens.data_ptr(), n_tokens.data_ptr(), score.data_ptr(), ws.data_ptr(),
                                        ws.numel() * 4, stream))
    fa = ws[:B * T].view(B, T)
    fp = ws[B * T:].view(torch.float32).view(B, T)
    return tokens, n_tokens, score, fa, fp


def _text(ids, vocabulary):
    return "".join(vocabulary[i] for i in ids).replace("<space>", " ")


def greedy_decoder(probs_seq, vocabulary, blank_index=0):
    """(score, text) of one [T, V] probability table.  ctc_greedy_decoder.py:6-31"""
    p = probs_seq if isinstance(probs_seq, torch.Tensor) else np.asarray(probs_seq, np.float32)
    if p.shape[0] == 0:
        return 0, ""
    tokens, n, score, _, _ = greedy_decode_ids(torch.as_tensor(p)[None], None, blank_index)
    n0 = int(n[0])
    s = float(score[0])
    return (s if s != 0.0 else 0), _text(tokens[0, :n0].tolist(), vocabulary)


def greedy_decoder_batch(probs_split, vocabulary, blank_index=0):
    """list[str] for a batch; like the reference, every row of every table is decoded (no length
    trimming, ctc_greedy_decoder.py:45-48).  Tables of equal length go to the GPU in one call."""
    if isinstance(probs_split, torch.Tensor) and probs_split.dim() == 3:
        tokens, n, _, _, _ = greedy_decode_ids(probs_split, None, blank_index)
        tk, nn = tokens.cpu(), n.cpu()
        return [_text(tk[b, :int(nn[b])].tolist(), vocabulary) for b in range(tk.shape[0])]
    return [greedy_decoder(p, vocabulary, blank_index)[1] for p in probs_split]


def greedy_decoder_chunk(probs_seq, vocabulary, last_max_prob_list=None, last_max_index_list=None, blank_index=0):
    """Streaming variant, ctc_greedy_decoder.py:52-89: the per-frame stage (argmax + max prob) runs on
    the GPU for the new chunk; the reference's stateful lists are kept with its (swapped) naming:
    ``last_max_prob_list`` holds argmax indices, ``last_max_index_list`` holds non-blank max probs."""
    if last_max_prob_list is None:
        last_max_prob_list = []
    if last_max_index_list is None:
        last_max_index_list = []
    p = probs_seq if isinstance(probs_seq, torch.Tensor) else np.asarray(probs_seq, np.float32)
    if p.shape[0] > 0:
        _, _, _, fa, fp = greedy_decode_ids(torch.as_tensor(p)[None], None, blank_index)
        idx = fa[0].cpu().numpy()
        prob = fp[0].cpu().numpy()
        last_max_prob_list.extend(list(idx))
        last_max_index_list.extend(list(prob[idx != blank_index]))
    hist = np.asarray(last_max_prob_list, np.int64)
    keep = np.ones(len(hist), bool)
    keep[1:] = hist[1:] != hist[:-1]
    ids = hist[keep]
    ids = ids[ids != blank_index]
    score = 0
    if len(last_max_index_list) > 0:
        score = float(sum(last_max_index_list) / len(last_max_index_list)) * 100.0
    return score, _text(ids, vocabulary), last_max_prob_list, last_max_index_list
