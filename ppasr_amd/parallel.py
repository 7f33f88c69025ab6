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

__all__ = ["shard_range", "pack_hypotheses", "unpack_hypotheses", "gather_hypotheses", "Bucket", "make_buckets",
           "assign_buckets", "ragged_record_shape", "gather_ragged_hypotheses"]


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of rank; remainders go to the lowest ranks."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_hypotheses(tokens, n_tokens, score):
    B, Tp = tokens.shape
    rec = torch.empty(B, Tp + 3, dtype=torch.int32, device=tokens.device)
    rec[:, :Tp] = tokens
    rec[:, Tp] = n_tokens
    rec[:, Tp + 1:] = score.to(torch.float64).contiguous().view(torch.int32).view(B, 2)
    return rec


def unpack_hypotheses(rec):
    Tp = rec.shape[1] - 3
    tokens = rec[:, :Tp]
    n_tokens = rec[:, Tp]
    score = rec[:, Tp + 1:].contiguous().view(torch.float64).view(-1)
    return tokens, n_tokens, score


def gather_hypotheses(tokens, n_tokens, score, dist, group=None):
    """All ranks end up with the hypotheses of the whole global batch, in rank order.
    Every rank must contribute the same [B_local, Tp] shape (pad the batch if needed)."""
    rec = pack_hypotheses(tokens, n_tokens, score)
    world = dist.get_world_size(group)
    out = torch.empty(world * rec.shape[0], rec.shape[1], dtype=torch.int32, device=rec.device)
    dist.all_gather_into_tensor(out, rec, group=group)
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


def gather_ragged_hypotheses(results, n_total, rows, cols, dist, device=None, group=None):
    """``results``: this rank's list of (utterance index, tokens 1-D int32 tensor, score float) in any order.
    -> (tokens [n_total, cols] i32 (-1 padded), n_tokens [n_total] i32, score [n_total] f64) in the CALLER's utterance
    order, identical on every rank.  One all-gather of ``int32[rows, cols + 4]`` per rank; unused rows carry index -1."""
    dev = device if device is not None else (results[0][1].device if results else "cpu")
    rec = torch.full((rows, cols + 4), -1, dtype=torch.int32, device=dev)
    if len(results) > rows:
        raise ValueError(f"{len(results)} results for a record of {rows} rows")
    for r, (idx, tok, sc) in enumerate(results):
        n = int(tok.numel())
        if n > cols:
            raise ValueError(f"hypothesis of {n} tokens for a record of {cols} columns")
        rec[r, :n] = tok.to(torch.int32)
        rec[r, cols] = n
        rec[r, cols + 1:cols + 3] = torch.tensor([float(sc)], dtype=torch.float64).view(torch.int32).to(dev)
        rec[r, cols + 3] = int(idx)
    world = dist.get_world_size(group)
    out = torch.empty(world * rows, cols + 4, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, rec, group=group)
    idx = out[:, cols + 3].to(torch.int64)
    valid = idx >= 0
    if int(valid.sum()) != n_total or sorted(idx[valid].tolist()) != list(range(n_total)):
        raise RuntimeError("ragged gather: the ranks' utterance indices do not partition the batch")
    order = torch.empty(n_total, dtype=torch.int64, device=dev)
    order[idx[valid]] = torch.nonzero(valid).flatten()
    g = out[order]
    score = g[:, cols + 1:cols + 3].contiguous().view(torch.float64).view(-1)
    return g[:, :cols], g[:, cols], score
