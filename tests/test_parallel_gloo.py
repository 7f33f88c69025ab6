"""world_size-2 gloo test of the multi-GPU path's only collective: the all-gather of packed
hypotheses (ppasr_amd/parallel.py), plus the shard arithmetic."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ppasr_amd.parallel import gather_hypotheses, pack_hypotheses, shard_range, unpack_hypotheses


def _make(rank, B, Tp):
    g = torch.Generator().manual_seed(100 + rank)
    tokens = torch.randint(1, 4233, (B, Tp), dtype=torch.int32, generator=g)
    n = torch.randint(0, Tp + 1, (B,), dtype=torch.int32, generator=g)
    for b in range(B):
        tokens[b, int(n[b]):] = -1
    score = torch.rand(B, dtype=torch.float64, generator=g) * 100
    return tokens, n, score


def _worker(rank, world, port, B, Tp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t, n, s = _make(rank, B, Tp)
    gt, gn, gs = gather_hypotheses(t, n, s, dist)
    ok = True
    for r in range(world):
        et, en, es = _make(r, B, Tp)
        ok &= bool((gt[r * B:(r + 1) * B] == et).all() and (gn[r * B:(r + 1) * B] == en).all()
                   and (gs[r * B:(r + 1) * B] == es).all())
    q.put((rank, ok))
    dist.destroy_process_group()


def test_pack_roundtrip_is_bit_exact():
    t, n, s = _make(0, 5, 17)
    a, b, c = unpack_hypotheses(pack_hypotheses(t, n, s))
    assert (a == t).all() and (b == n).all() and (c == s).all()


def test_shard_ranges_partition_the_batch():
    for n in (1, 7, 32, 33, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_all_gather_world2_gloo():
    world, B, Tp = 2, 4, 33
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, Tp, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res
