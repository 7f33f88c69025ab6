"""Utterance-level data parallelism for inference: one process per GPU,
weights replicated, and ONE all-gather (RCCL over xGMI with backend "nccl"; gloo in the CPU
tests) of a fixed-size packed hypothesis record per rank.

* equal-length batches (BASELINE configs[2]): the batch is sharded contiguously by rank
  (``shard_range``) and ``gather_hypotheses`` returns the global batch in rank order;
* variable-length batches (configs[4]): utterances are cut into length buckets (200-frame width, a
  bucket is padded to its longest member) and WHOLE buckets are dealt to the ranks by greedy
  bin-packing in order of descending padded work (``assign_buckets``), so that the ranks finish
  together; ``gather_ragged_hypotheses`` restores the caller's utterance order after the gather.
  Every rank knows every length, so the plan and the record shape need no collective of their own.

The reference has no multi-GPU inference path (its only collective is training DP,
trainer.py:529-544); this is the north-star's new capability, not a port.

Record layout per utterance (int32 words): tokens[L] (-1 padded) | n_tokens | score (f64 as 2 words)
[| original utterance index, ragged route only].
"""
import torch

from ppasr_amd import _lib

__all__ = ["shard_range", "pack_hypotheses", "unpack_hypotheses", "gather_hypotheses", "Bucket", "make_buckets",
           "assign_buckets", "ragged_record_shape", "gather_ragged_hypotheses", "rank_batches", "greedy_ids_decoder",
           "beam_ids_decoder", "decode_ragged", "RaggedPlan"]


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of rank; remainders go to the lowest ranks."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _hip_records(*tensors):
    """Device records are packed / unpacked by the library's own kernels (ppasr_hyp_pack / ppasr_hyp_unpack: one launch
    each); host tensors (the gloo tests of the plan and gather logic) by the plain tensor expressions below."""
    return all(t.is_cuda for t in tensors)


def _hip_pack(tokens, n_tokens, score, index, rec, row0, cols, extra):
    lib = _lib.load()
    tokens = tokens if (tokens.dtype == torch.int32 and tokens.stride(1) == 1) else tokens.to(torch.int32).contiguous()
    n_tokens = n_tokens if n_tokens.dtype == torch.int32 else n_tokens.to(torch.int32)
    score = score if score.dtype == torch.float64 else score.to(torch.float64)
    k, L = tokens.shape
    with torch.cuda.device(rec.device):
        _lib.check(lib.ppasr_hyp_pack(tokens.data_ptr(), tokens.stride(0), L, n_tokens.data_ptr(), n_tokens.stride(0),
                                      score.data_ptr(), score.stride(0), None if index is None else index.data_ptr(), k,
                                      rec.data_ptr(), row0, cols, extra, torch.cuda.current_stream(rec.device).cuda_stream))


def _hip_unpack(rec, order, n_rows, cols, extra, want_index=False):
    lib = _lib.load()
    dev = rec.device
    tokens = torch.empty(n_rows, cols, dtype=torch.int32, device=dev)
    n_tokens = torch.empty(n_rows, dtype=torch.int32, device=dev)
    score = torch.empty(n_rows, dtype=torch.float64, device=dev)
    index = torch.empty(n_rows, dtype=torch.int32, device=dev) if want_index else None
    with torch.cuda.device(dev):
        _lib.check(lib.ppasr_hyp_unpack(rec.data_ptr(), None if order is None else order.data_ptr(), n_rows, cols, extra,
                                        tokens.data_ptr(), n_tokens.data_ptr(), score.data_ptr(),
                                        None if index is None else index.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    return tokens, n_tokens, score, index


def pack_hypotheses(tokens, n_tokens, score):
    B, Tp = tokens.shape
    rec = torch.empty(B, Tp + 3, dtype=torch.int32, device=tokens.device)
    if _hip_records(tokens, n_tokens, score):
        _hip_pack(tokens, n_tokens, score, None, rec, 0, Tp, 3)
        return rec
    rec[:, :Tp] = tokens
    rec[:, Tp] = n_tokens
    rec[:, Tp + 1:] = score.to(torch.float64).contiguous().view(torch.int32).view(B, 2)
    return rec


def unpack_hypotheses(rec):
    Tp = rec.shape[1] - 3
    if rec.is_cuda:
        tokens, n_tokens, score, _ = _hip_unpack(rec, None, rec.shape[0], Tp, 3)
        return tokens, n_tokens, score
    tokens = rec[:, :Tp]
    n_tokens = rec[:, Tp]
    score = rec[:, Tp + 1:].contiguous().view(torch.float64).view(-1)
    return tokens, n_tokens, score


def _all_gather_rows(out, rec, dist, group=None):
    """``dist.all_gather_into_tensor(out, rec)``; device tensors under a host-only backend (gloo: the shared-GPU plumbing
    check of ``bench.py --share-gpu``, CPU tests) are staged through the host -- RCCL takes them as they are."""
    try:
        backend = str(dist.get_backend(group))
    except Exception:  # a stub "dist" in tests
        backend = "gloo"
    if rec.is_cuda and backend != "nccl":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, rec.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, rec, group=group)


def gather_hypotheses(tokens, n_tokens, score, dist, group=None):
    """All ranks end up with the hypotheses of the whole global batch, in rank order.
    Every rank must contribute the same [B_local, Tp] shape (pad the batch if needed)."""
    rec = pack_hypotheses(tokens, n_tokens, score)
    world = dist.get_world_size(group)
    out = torch.empty(world * rec.shape[0], rec.shape[1], dtype=torch.int32, device=rec.device)
    _all_gather_rows(out, rec, dist, group)
    return unpack_hypotheses(out)


# ---- variable-length batches: whole length buckets per rank --------------------------------------------------------
class Bucket:
    """Utterances ``indices`` (positions in the caller's batch) padded to ``frames`` input frames."""

    __slots__ = ("indices", "frames", "cost")

    def __init__(self, indices, frames):
        self.indices = list(indices)
        self.frames = int(frames)
        self.cost = len(self.indices) * self.frames  # padded frames the encoder computes for this bucket

    def __repr__(self):
        return f"Bucket(n={len(self.indices)}, frames={self.frames})"


def make_buckets(lengths, width=200):
    """Length buckets of ``width`` frames ((len-1)//width), each padded to its longest member
    (SURVEY.md §8d cfg5).  Buckets come back longest first; indices inside a bucket keep the caller's order."""
    groups = {}
    for i, ln in enumerate(lengths):
        ln = int(ln)
        if ln <= 0:
            raise ValueError("utterance lengths must be positive")
        groups.setdefault((ln - 1) // width, []).append(i)
    out = [Bucket(idx, max(int(lengths[i]) for i in idx)) for _, idx in sorted(groups.items(), reverse=True)]
    return out


def assign_buckets(lengths, world, width=200):
    """-> list over ranks of lists of ``Bucket``: whole buckets, taken in order of descending padded work
    (utterances x padded frames) and given to the least loaded rank so far (greedy LPT bin-packing; ties go to the
    lowest rank, so every rank computes the same plan).  A rank may get no bucket when there are fewer buckets than
    ranks."""
    if world < 1:
        raise ValueError("world must be >= 1")
    buckets = sorted(make_buckets(lengths, width), key=lambda b: (-b.cost, -b.frames, b.indices[0]))
    plan = [[] for _ in range(world)]
    load = [0] * world
    for b in buckets:
        r = min(range(world), key=lambda k: (load[k], k))
        plan[r].append(b)
        load[r] += b.cost
    return plan


def ragged_record_shape(lengths, world, out_frames, width=200):
    """(rows, token_columns) of the per-rank record every rank allocates: the largest utterance count of any rank
    and the output frames of the longest utterance (``out_frames``: input frames -> output frames of the model)."""
    plan = assign_buckets(lengths, world, width)
    rows = max(1, max(sum(len(b.indices) for b in p) for p in plan))
    cols = max(1, int(out_frames(max(int(v) for v in lengths))))
    return rows, cols


def set_skip_padding_if_built(model, enable):
    """``model.set_skip_padding(enable)`` where the model's route has the ragged mode; the routes that compute every row
    (general layer route, the 6x / 8x front ends, DeepSpeech2 -- the library answers PPASR_EUNSUPPORTED) keep doing so:
    the decoders trim by ``frame_lens`` either way, only the padded rows' compute is not saved.  -> whether it is on."""
    fn = getattr(model, "set_skip_padding", None)
    if fn is None:
        return False
    try:
        fn(enable)
        return bool(enable)
    except _lib.PPASRHipError as e:
        if e.status == _lib.PPASR_EUNSUPPORTED:  # (any other status -- an allocation failure, a bad handle -- is an error)
            return False
        raise


def _collective_device(dist, device, group=None):
    """Device the collective's tensors must live on: the caller's choice, else the current HIP device under the
    nccl (= RCCL) backend (a rank that was dealt no bucket has no result tensor to take it from), else the host."""
    if device is not None:
        return torch.device(device)
    try:
        backend = dist.get_backend(group)
    except Exception:  # a stub "dist" in tests
        backend = "gloo"
    if str(backend) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_ragged_hypotheses(results, n_total, rows, cols, dist, device=None, group=None):
    """``results``: this rank's list of (utterance index, tokens 1-D int32 tensor / array, score float) in any order.
    -> (tokens [n_total, cols] i32 (-1 padded), n_tokens [n_total] i32, score [n_total] f64) in the CALLER's utterance
    order, identical on every rank.  One all-gather of ``int32[rows, cols + 4]`` per rank; unused rows carry index -1.
    The record is assembled on the host and uploaded once."""
    import numpy as np
    dev = _collective_device(dist, device, group)
    if len(results) > rows:
        raise ValueError(f"{len(results)} results for a record of {rows} rows")
    host = np.full((rows, cols + 4), -1, dtype=np.int32)
    for r, (idx, tok, sc) in enumerate(results):
        tok = tok.detach().cpu().numpy() if isinstance(tok, torch.Tensor) else np.asarray(tok)
        n = int(tok.size)
        if n > cols:
            raise ValueError(f"hypothesis of {n} tokens for a record of {cols} columns")
        host[r, :n] = tok.astype(np.int32, copy=False)
        host[r, cols] = n
        host[r, cols + 1:cols + 3] = np.array([float(sc)], dtype=np.float64).view(np.int32)
        host[r, cols + 3] = int(idx)
    rec = torch.from_numpy(host).to(dev)
    world = dist.get_world_size(group)
    out = torch.empty(world * rows, cols + 4, dtype=torch.int32, device=dev)
    _all_gather_rows(out, rec, dist, group)
    idx = out[:, cols + 3].to(torch.int64)
    valid = idx >= 0
    if int(valid.sum()) != n_total or sorted(idx[valid].tolist()) != list(range(n_total)):
        raise RuntimeError("ragged gather: the ranks' utterance indices do not partition the batch")
    order = torch.empty(n_total, dtype=torch.int64, device=dev)
    order[idx[valid]] = torch.nonzero(valid).flatten()
    g = out[order]
    score = g[:, cols + 1:cols + 3].contiguous().view(torch.float64).view(-1)
    return g[:, :cols], g[:, cols], score


# ---- end to end: plan -> encode -> decode -> gather ---------------------------------------------------------------------
def rank_batches(lengths, rank, world, width=200, mode="merged", max_batch_frames=None):
    """The encoder batches of ``rank``: lists of utterance indices (positions in the caller's batch).
    ``assign_buckets`` deals WHOLE 200-frame buckets to the ranks; then
      mode "buckets": one batch per bucket (padded to the bucket's longest member; every padded row is computed -- the
                      reference's arithmetic for that batch composition);
      mode "merged":  the rank's buckets merged into ONE batch, longest utterance first, to be run with the encoder's
                      ragged mode (``set_skip_padding``: rows behind an utterance's valid frames are not computed, so
                      the padding costs nothing, while the merged grid fills the chip -- a bucket of 1-3 utterances is
                      24-70 row blocks on 256 CUs).  ``max_batch_frames`` (sum of padded frames) cuts the merged list
                      into several batches when a rank's share is larger than one workspace should hold."""
    mine = assign_buckets(lengths, world, width)[rank]
    if mode == "buckets":
        return [list(b.indices) for b in sorted(mine, key=lambda b: -b.frames)]
    if mode != "merged":
        raise ValueError(f"mode {mode!r}: 'merged' or 'buckets'")
    idx = sorted((i for b in mine for i in b.indices), key=lambda i: (-int(lengths[i]), i))
    if not idx:
        return []
    if not max_batch_frames:
        return [idx]
    out, cur = [], []
    for i in idx:
        tmax = int(lengths[cur[0]]) if cur else int(lengths[i])
        if cur and (len(cur) + 1) * tmax > max_batch_frames:
            out.append(cur)
            cur = []
        cur.append(i)
    out.append(cur)
    return out


def greedy_ids_decoder(blank=0):
    """decoder callable for ``decode_ragged``: CTC greedy over each utterance's valid frames."""
    from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decode_ids

    def run(probs, frame_lens):
        tokens, n, score, _, _ = greedy_decode_ids(probs, frame_lens, blank)
        return tokens, n, score
    return run


def beam_ids_decoder(beam_size, cutoff_prob=1.0, cutoff_top_n=40, blank=0, ext_scorer=None):
    """decoder callable for ``decode_ragged``: CTC prefix beam search (best hypothesis) over the valid frames."""
    from ppasr_amd.decoders.beam_search_decoder import beam_search_ids

    def run(probs, frame_lens):
        tokens, n, score, _ = beam_search_ids(probs, beam_size, cutoff_prob, cutoff_top_n, blank, frame_lens=frame_lens,
                                              nbest=1, ext_scorer=ext_scorer)
        return tokens[:, 0], n[:, 0], score[:, 0]
    return run


def _pad_batch(feats, lengths, idx, device):
    """Zero-padded [n, Tmax, F] batch (collate_fn.py:17) + lengths of the listed utterances.  ``feats``: a padded
    [N, T, F] tensor / array, or a sequence of per-utterance [T_i, F] arrays (only this rank's are touched)."""
    tmax = max(int(lengths[i]) for i in idx)
    lens = torch.tensor([int(lengths[i]) for i in idx], dtype=torch.int64)
    if isinstance(feats, torch.Tensor) and feats.dim() == 3:
        x = feats[torch.as_tensor(idx, dtype=torch.int64, device=feats.device), :tmax]
        return x.to(device).contiguous(), lens.to(device)
    first = torch.as_tensor(feats[idx[0]])
    x = torch.zeros(len(idx), tmax, first.shape[-1], dtype=torch.float32)
    for j, i in enumerate(idx):
        f = torch.as_tensor(feats[i], dtype=torch.float32)
        n = int(lengths[i])
        x[j, :n] = f[:n]
    return x.to(device), lens.to(device)


class RaggedPlan:
    """A variable-length batch prepared for repeated / pipelined decoding on this rank: the plan (``assign_buckets``),
    this rank's encoder batches resident on the device, and the record geometry of the gather.

    ``run(decoder)`` = encode -> decode -> (gather) of the prepared batch; with ``pipeline=True`` the encoder runs on
    its own HIP stream and the decoder (+ gather) on a second one, linked by events only, so that consecutive ``run``
    calls overlap: the beam search of call i (one workgroup per utterance, latency-bound) runs while the encoder of
    call i+1 has the rest of the chip.  Results of every call are complete once its decode stream is synchronised."""

    def __init__(self, model, feats, lengths, dist=None, group=None, width=200, mode="merged", max_batch_frames=None,
                 device=None, pipeline=False):
        self.model = model
        self.lengths = [int(v) for v in lengths]
        self.n_total = len(self.lengths)
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.rank = dist.get_rank(group) if dist is not None else 0
        self.dev = torch.device(device) if device is not None else torch.device(getattr(model, "device", "cpu"))
        self.ragged = mode == "merged"
        self.mode = mode
        plan = assign_buckets(self.lengths, self.world, width)
        self.rows = max(1, max(sum(len(b.indices) for b in p) for p in plan))
        self.cols = max(1, int(model.out_frames(max(self.lengths))))
        # row of every utterance in the gathered record: every rank's batches are known from the lengths alone, so the
        # caller's order is restored with one precomputed index (no data-dependent step, no host synchronisation)
        order = [-1] * self.n_total
        for r in range(self.world):
            pos = 0
            for idx in rank_batches(self.lengths, r, self.world, width, mode, max_batch_frames):
                for i in idx:
                    if order[i] != -1:
                        raise RuntimeError("ragged plan: an utterance was dealt to two ranks")
                    order[i] = r * self.rows + pos
                    pos += 1
        if any(o < 0 for o in order):
            raise RuntimeError("ragged plan: the ranks' batches do not cover the batch")
        self.order = torch.tensor(order, dtype=torch.int64, device=self.dev)
        self.batches = []
        # the plan knows every length on the host: each ragged batch's lengths go to models that can use them to pick kernel
        # variants (ppasr_set_lengths_hint)
        self.hints = []
        for idx in rank_batches(self.lengths, self.rank, self.world, width, mode, max_batch_frames):
            x, lens = _pad_batch(feats, self.lengths, idx, self.dev)
            frame_lens = model.valid_out_frames(lens, x.shape[1])
            self.batches.append((torch.tensor(idx, dtype=torch.int32, device=self.dev), x, lens, frame_lens))
            self.hints.append([int(self.lengths[i]) for i in idx] if hasattr(model, "set_lengths_hint") else None)
        self.cuda = self.dev.type == "cuda"
        self._rec = self._gathered = None  # device record buffers, kept across runs (see _record)
        self.pipeline = bool(pipeline) and self.cuda
        # (plain streams: the search's workgroups and the encoder's share the chip.  Giving each side its own CUs through
        #  hipExtStreamCreateWithCUMask was measured -- tools/cu_mask_probe.hip, NOTES.md 9.5 -- and is slower: 8.9 ms per
        #  cfg5 step against 6.8, wherever the search's CUs are placed)
        self.enc_stream = torch.cuda.Stream(device=self.dev) if self.pipeline else None
        self.dec_stream = torch.cuda.Stream(device=self.dev) if self.pipeline else None
        # the batches above were prepared (gather, pad, .contiguous()) on the CALLER's current stream: the side streams
        # must not read them before that work is done
        self.ready = None
        if self.pipeline:
            self.ready = torch.cuda.Event()
            self.ready.record(torch.cuda.current_stream(self.dev))
            for (idx, x, lens, frame_lens) in self.batches:
                for t in (idx, x, lens, frame_lens):
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(self.enc_stream)
                        t.record_stream(self.dec_stream)

    def _record(self, outs):
        """int32 [rows, cols + 4] record of this rank: tokens (-1 padded) | n_tokens | score (f64 as two words) |
        utterance index (-1: unused row).  On the device: ONE launch of the library's packing kernel per decoder call into
        a record buffer that is kept across runs (rows this rank never uses were set to -1 once; the buffer is only touched
        in stream order -- pack, gather, unpack of run i come before the pack of run i + 1 on the same stream)."""
        cols = self.cols
        if self.cuda and all(t.is_cuda for o in outs for t in o):
            if self._rec is None:
                self._rec = torch.full((self.rows, cols + 4), -1, dtype=torch.int32, device=self.dev)
            r = 0
            for (idx, _x, _l, _fl), (tokens, n, score) in zip(self.batches, outs):
                if int(tokens.shape[1]) > cols and bool((n > cols).any()):
                    raise ValueError(f"a hypothesis is longer than the record's {cols} columns")
                _hip_pack(tokens, n, score, idx, self._rec, r, cols, 4)  # (columns beyond `cols` are not read)
                r += tokens.shape[0]
            return self._rec
        rec = torch.full((self.rows, cols + 4), -1, dtype=torch.int32, device=self.dev)
        r = 0
        for (idx, _x, _l, _fl), (tokens, n, score) in zip(self.batches, outs):
            k = tokens.shape[0]
            w = min(int(tokens.shape[1]), cols)
            if int(tokens.shape[1]) > cols and bool((n > cols).any()):
                raise ValueError(f"a hypothesis is longer than the record's {cols} columns")
            rec[r:r + k, :w] = tokens[:, :w].to(torch.int32)
            rec[r:r + k, cols] = n.to(torch.int32)
            rec[r:r + k, cols + 1:cols + 3] = score.to(torch.float64).contiguous().view(torch.int32).view(k, 2)
            rec[r:r + k, cols + 3] = idx
            r += k
        return rec

    def _unpack(self, out):
        cols = self.cols
        if out.is_cuda:
            tokens, n, score, self.last_index_column = _hip_unpack(out, self.order, self.n_total, cols, 4, want_index=True)
            return tokens, n, score
        g = out[self.order.to(out.device)]
        score = g[:, cols + 1:cols + 3].contiguous().view(torch.float64).view(-1)
        return g[:, :cols], g[:, cols], score

    def verify(self, gathered_index_column):
        """Host-side check (synchronises): the record rows really carry the utterance indices the plan expects."""
        return bool((gathered_index_column.cpu() == torch.arange(self.n_total, dtype=torch.int32)).all())

    def _encode_decode(self, decoder):
        model = self.model
        outs = []
        if self.ragged and self.batches:
            set_skip_padding_if_built(model, True)
        prev_front = getattr(model, "front_fused", -1)
        if self.pipeline and hasattr(model, "set_front_fused"):
            # next to the previous call's beam search the two-launch front end is the faster one (see the C header); for
            # this call only: the caller's own setting is restored below
            model.set_front_fused(0)
        if self.pipeline:
            # inputs ready (see __init__).  enc_stream does NOT wait for the previous call's decode: that is the overlap --
            # call i's decoder reads only probs(i) (its own allocation, record_stream below) and the decoder's state, the
            # encoder of call i + 1 only the model's workspace and these inputs, all in stream order on enc_stream
            self.enc_stream.wait_event(self.ready)
            self.dec_stream.wait_event(self.ready)
        try:
            for (_idx, x, lens, frame_lens), hint in zip(self.batches, self.hints):
                if hint is not None:
                    model.set_lengths_hint(hint)  # (host-side route selection for the ragged batch; see the C header)
                if self.pipeline:
                    with torch.cuda.stream(self.enc_stream):
                        probs = model.get_encoder_out(x, lens)
                        ev = torch.cuda.Event()
                        ev.record(self.enc_stream)
                    with torch.cuda.stream(self.dec_stream):
                        self.dec_stream.wait_event(ev)
                        probs.record_stream(self.dec_stream)
                        outs.append(decoder(probs, frame_lens))
                else:
                    probs = model.get_encoder_out(x, lens)
                    outs.append(decoder(probs, frame_lens))
        finally:
            if self.ragged and self.batches:
                set_skip_padding_if_built(model, False)
            if any(h is not None for h in self.hints):
                model.set_lengths_hint(None)
            if self.pipeline and hasattr(model, "set_front_fused"):
                model.set_front_fused(prev_front)
        return outs

    def run(self, decoder):
        """-> (tokens [N, L] i32 -1 padded, n_tokens [N] i32, score [N] f64) device tensors in the caller's utterance
        order, identical on every rank.  With ``pipeline`` the tensors are produced on ``self.dec_stream``: synchronise
        it (or make your stream wait for it) before reading them."""
        outs = self._encode_decode(decoder)
        ctx = torch.cuda.stream(self.dec_stream) if self.pipeline else _null_ctx()
        with ctx:
            rec = self._record(outs)
            if self.dist is None:
                out = rec
            else:
                if self._gathered is None or self._gathered.device != rec.device:
                    self._gathered = torch.empty(self.world * self.rows, self.cols + 4, dtype=torch.int32, device=rec.device)
                out = self._gathered
                _all_gather_rows(out, rec, self.dist, self.group)
            if not out.is_cuda:
                self.last_index_column = out[self.order.to(out.device), self.cols + 3]
            return self._unpack(out)

    def sync(self):
        if self.pipeline:
            self.dec_stream.synchronize()
            self.enc_stream.synchronize()


class _null_ctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def decode_ragged(model, feats, lengths, decoder, dist=None, group=None, width=200, mode="merged", max_batch_frames=None,
                  device=None):
    """Variable-length batch, end to end (BASELINE configs[4]): plan (``assign_buckets``: whole length buckets per rank)
    -> this rank's encoder batches (``rank_batches``) -> ``model.get_encoder_out`` -> ``decoder(probs, frame_lens)`` ->
    ONE all-gather of the packed hypotheses.

    model:   a ppasr_amd model (``get_encoder_out``, ``valid_out_frames``, ``out_frames``, ``set_skip_padding``);
    feats:   padded [N, T, F] tensor / array or a sequence of per-utterance [T_i, F] arrays, caller's order (only this
             rank's utterances are touched);
    lengths: [N] input frames of every utterance (every rank knows all of them: the plan needs no collective);
    decoder: ``greedy_ids_decoder()`` / ``beam_ids_decoder(...)`` or any callable (probs [n,T',V], frame_lens [n] i32)
             -> (tokens [n,L] i32 -1 padded, n_tokens [n] i32, score [n] f64) device tensors;
    dist:    an initialised ``torch.distributed`` (or None: one rank);
    mode:    "merged" (one ragged batch per rank, ``set_skip_padding``) or "buckets" (one padded batch per bucket).
    -> (tokens [N, L] i32, n_tokens [N] i32, score [N] f64) in the caller's order, identical on every rank, where
    L = out_frames(longest utterance).  Decoding covers each utterance's VALID output frames only.
    Repeated or pipelined decoding of prepared batches: ``RaggedPlan``."""
    plan = RaggedPlan(model, feats, lengths, dist=dist, group=group, width=width, mode=mode,
                      max_batch_frames=max_batch_frames, device=device)
    tokens, n, score = plan.run(decoder)
    if not plan.verify(plan.last_index_column):
        raise RuntimeError("ragged gather: the ranks' utterance indices do not partition the batch")
    return tokens, n, score
