"""Utterance-level data parallelism for inference: one process per GPU,
batch sharded contiguously by rank, weights replicated, and ONE all-gather (RCCL over xGMI
with backend "nccl"; gloo in the CPU tests) of a fixed-size packed hypothesis record per rank.

The reference has no multi-GPU inference path (its only collective is training DP,
trainer.py:529-544); this is the north-star's new capability, not a port.

Record layout per utterance (int32 words): tokens[Tp] (-1 padded) | n_tokens | score (f64 as 2 words).
"""
import torch

__all__ = ["shard_range", "pack_hypotheses", "unpack_hypotheses", "gather_hypotheses"]


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
