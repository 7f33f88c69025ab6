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


# ---- variable-length batches: whole length buckets per rank (BASELINE configs[4]) ------------------------
import numpy as np  # noqa: E402

from ppasr_amd.parallel import (assign_buckets, gather_ragged_hypotheses, make_buckets,  # noqa: E402
                                ragged_record_shape)


def _cfg5_lengths(n=128, seed=20740):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(200, 3001, size=n)


def test_buckets_partition_and_pad_to_their_longest_member():
    lens = _cfg5_lengths()
    buckets = make_buckets(lens, 200)
    seen = sorted(i for b in buckets for i in b.indices)
    assert seen == list(range(len(lens)))
    for b in buckets:
        ls = [int(lens[i]) for i in b.indices]
        assert b.frames == max(ls)
        assert len({(l - 1) // 200 for l in ls}) == 1          # one 200-frame class per bucket
        assert b.cost == len(ls) * b.frames
    assert [b.frames for b in buckets] == sorted((b.frames for b in buckets), reverse=True)


def test_assign_buckets_is_balanced_and_deterministic():
    lens = _cfg5_lengths()
    for world in (1, 2, 4, 8):
        plan = assign_buckets(lens, world)
        assert plan == plan and len(plan) == world
        got = sorted(i for p in plan for b in p for i in b.indices)
        assert got == list(range(len(lens)))                      # whole buckets, every utterance exactly once
        load = [sum(b.cost for b in p) for p in plan]
        biggest = max(b.cost for p in plan for b in p)
        # greedy LPT: no rank exceeds the mean by more than the largest single bucket
        assert max(load) - sum(load) / world <= biggest
        if world == 8:
            assert max(load) / (sum(load) / world) < 1.15, load   # 128 utterances in 15 buckets over 8 ranks
        again = assign_buckets(list(lens), world)
        assert [[b.indices for b in p] for p in again] == [[b.indices for b in p] for p in plan]
    # fewer buckets than ranks: the surplus ranks get nothing, nothing is lost
    plan = assign_buckets([250, 260, 900], 4)
    assert sorted(len(p) for p in plan) == [0, 0, 1, 1]


def _stub_decode(index, length):
    """Stands in for encoder + decoder on one utterance: a token sequence and score that depend on the utterance only."""
    g = torch.Generator().manual_seed(7000 + index)
    n = int(length) // 40
    return torch.randint(1, 4233, (n,), dtype=torch.int32, generator=g), float(index) * 0.5 + n


def _ragged_worker(rank, world, port, lens, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out_frames = lambda T: ((T - 1) // 2 - 1) // 2
    rows, cols = ragged_record_shape(lens, world, out_frames)
    mine = assign_buckets(lens, world)[rank]
    results = []
    for b in mine:                       # one "batch" per bucket, like the GPU path
        for i in b.indices:
            tok, sc = _stub_decode(i, lens[i])
            results.append((i, tok, sc))
    tokens, n, score = gather_ragged_hypotheses(results, len(lens), rows, cols, dist)
    ok = True
    for i, ln in enumerate(lens):        # the caller's utterance order is restored on every rank
        tok, sc = _stub_decode(i, ln)
        ok &= int(n[i]) == tok.numel() and bool((tokens[i, :tok.numel()] == tok).all()) and float(score[i]) == sc
        ok &= bool((tokens[i, tok.numel():] == -1).all())
    q.put((rank, ok, len(results)))
    dist.destroy_process_group()


def test_ragged_shard_decode_gather_restores_order_world2_gloo():
    world = 2
    lens = [int(v) for v in _cfg5_lengths(16, 20741)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, lens, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert sum(n for _, _, n in res) == len(lens)


# ---- evaluate(): batches dealt round-robin to the ranks, one all-reduce of (sum, count) -----------------------------
def _eval_batches():
    vocab = ["<blank>", "<unk>"] + [chr(0x4E00 + i) for i in range(20)] + ["<eos>"]
    rng = np.random.Generator(np.random.PCG64(5))
    batches = []
    for k in range(5):
        B, U = 3, 6
        labels = rng.integers(2, 22, size=(B, U)).astype(np.int64)
        # "recognised" ids: the labels with one substitution in utterance k % B  -> known error rates
        hyp = labels.copy()
        hyp[k % B, 0] = 2 + (hyp[k % B, 0] - 1) % 20
        batches.append((hyp, labels, np.full(B, 40, np.int64), np.full(B, U, np.int64)))
    return vocab, batches


class _StubModel:
    device = None

    def get_encoder_out(self, inputs, input_lens):
        return inputs                    # the "probabilities" are the recognised ids; decoding is stubbed below


def _eval_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import ppasr_amd.evaluate as ev
    vocab, batches = _eval_batches()
    ev.decoder_result = lambda outs, vocabulary, *a, **k: ["".join(vocabulary[i] for i in row) for row in outs.tolist()]
    val = ev.evaluate(_StubModel(), batches, vocab, decoder="ctc_greedy", metrics_type="cer", overlap_decode=False)
    q.put((rank, val))
    if world > 1:
        dist.destroy_process_group()


def test_evaluate_all_reduce_world2_gloo_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_eval_worker, args=(0, 1, 0, q))
    p.start()
    _, single = q.get(timeout=120)
    p.join(timeout=60)
    # 5 batches x 3 utterances, one substitution in 6 characters in one utterance per batch
    assert abs(single - (5 * (1 / 6)) / 15) < 1e-12
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert all(abs(v - single) < 1e-12 for _, v in res), (single, res)


# ---- decode_ragged: plan -> encode -> decode -> gather, with a CPU model that has the real model interface ----------
class _CpuRaggedModel:
    """The interface ``decode_ragged`` drives (get_encoder_out / valid_out_frames / out_frames / set_skip_padding /
    device) on the CPU: output frame t of an utterance is a one-hot "probability" row whose class is read from the
    utterance's own feature frame 4t, so a hypothesis depends on the utterance only, not on what it was batched with."""
    device = torch.device("cpu")
    V = 50

    def __init__(self):
        self.calls = []          # (batch size, padded frames, skip_padding) of every encode
        self.skip = False

    def out_frames(self, T):
        return ((int(T) - 1) // 2 - 1) // 2

    def valid_out_frames(self, lens, T):
        # (the real models count frame t valid iff 4t < len, capped by the BATCH's output frames; the stub uses the
        #  utterance's own output frames so that the expected hypothesis does not depend on the batch it lands in)
        lens = torch.as_tensor(lens, dtype=torch.int64)
        return torch.clamp(((lens - 1) // 2 - 1) // 2, min=0, max=self.out_frames(T)).to(torch.int32)

    def set_skip_padding(self, enable=True):
        self.skip = bool(enable)

    def get_encoder_out(self, x, lens):
        B, T, _ = x.shape
        Tp = self.out_frames(T)
        self.calls.append((B, T, self.skip))
        cls = (x[:, 0:4 * Tp:4, 0].round().to(torch.int64) % self.V)
        return torch.nn.functional.one_hot(cls, self.V).to(torch.float32)


def _cpu_greedy(probs, frame_lens):
    B, Tp, _ = probs.shape
    ids = probs.argmax(-1)
    tokens = torch.full((B, Tp), -1, dtype=torch.int32)
    n = torch.zeros(B, dtype=torch.int32)
    score = torch.zeros(B, dtype=torch.float64)
    for b in range(B):
        row = ids[b, :int(frame_lens[b])].tolist()
        out = [c for j, c in enumerate(row) if c != 0 and (j == 0 or c != row[j - 1])]
        tokens[b, :len(out)] = torch.tensor(out, dtype=torch.int32)
        n[b] = len(out)
        score[b] = float(len(row))
    return tokens, n, score


def _ragged_feats(lens, seed=11):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, 50, (int(l), 3), generator=g).to(torch.float32) for l in lens]


def _decode_ragged_worker(rank, world, port, lens, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ppasr_amd.parallel import decode_ragged
    model = _CpuRaggedModel()
    feats = _ragged_feats(lens)
    tokens, n, score = decode_ragged(model, feats, lens, _cpu_greedy, dist=dist, mode=mode)
    single = _CpuRaggedModel()
    ok = True
    for i, ln in enumerate(lens):       # expected: the utterance decoded on its own
        p = single.get_encoder_out(feats[i][None], [ln])
        et, en, es = _cpu_greedy(p, single.valid_out_frames([ln], ln))
        ok &= int(n[i]) == int(en[0]) and bool((tokens[i, :int(en[0])] == et[0, :int(en[0])]).all())
        ok &= bool((tokens[i, int(en[0]):] == -1).all()) and float(score[i]) == float(es[0])
    q.put((rank, ok, model.calls))
    dist.destroy_process_group()


def test_decode_ragged_world2_gloo_both_modes():
    from ppasr_amd.parallel import assign_buckets
    world = 2
    lens = [int(v) for v in _cfg5_lengths(16, 20741)]
    ctx = mp.get_context("spawn")
    for k, mode in enumerate(("merged", "buckets")):
        q = ctx.Queue()
        port = 35500 + os.getpid() % 2000 + k
        procs = [ctx.Process(target=_decode_ragged_worker, args=(r, world, port, lens, mode, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
        assert all(ok for _, ok, _ in res), (mode, res)
        plan = assign_buckets(lens, world)
        for rank, _, calls in res:
            if mode == "merged":    # ONE ragged batch per rank, padded to the rank's longest utterance, skip_padding on
                assert len(calls) == 1 and calls[0][2] is True
                assert calls[0][0] == sum(len(b.indices) for b in plan[rank])
                assert calls[0][1] == max(b.frames for b in plan[rank])
            else:                   # one batch per bucket, every padded row computed
                assert sorted((c[0], c[1]) for c in calls) == sorted((len(b.indices), b.frames) for b in plan[rank])
                assert not any(c[2] for c in calls)


def test_decode_ragged_single_process_and_rank_without_bucket():
    """No process group: the whole batch on one rank; and a plan that leaves a rank empty still gathers (the empty rank
    contributes a record of -1 rows on the collective's device)."""
    from ppasr_amd.parallel import decode_ragged, rank_batches
    lens = [250, 260, 900]
    model = _CpuRaggedModel()
    feats = _ragged_feats(lens, 5)
    tokens, n, score = decode_ragged(model, feats, lens, _cpu_greedy)
    assert tokens.shape == (3, model.out_frames(900)) and int((n > 0).sum()) == 3
    assert rank_batches(lens, 3, 4) == [] and rank_batches(lens, 2, 4) == []
    assert rank_batches(lens, 0, 1, max_batch_frames=1000) == [[2], [1, 0]]


def test_evaluate_bucket_sharding_equals_plain_evaluate():
    import ppasr_amd.evaluate as ev
    vocab = ["<blank>"] + [chr(0x4E00 + i) for i in range(49)] + ["<eos>"]   # ids 0..49 are the model's classes
    lens = [250, 260, 900, 1300, 420]
    feats = _ragged_feats(lens, 9)
    T = max(lens)
    x = torch.zeros(len(lens), T, 3)
    for i, f in enumerate(feats):
        x[i, :lens[i]] = f
    model = _CpuRaggedModel()
    labels = np.full((len(lens), 400), -1, np.int64)
    for i, ln in enumerate(lens):
        et, en, _ = _cpu_greedy(model.get_encoder_out(feats[i][None], [ln]), model.valid_out_frames([ln], ln))
        ids = et[0, :int(en[0])].numpy()
        labels[i, :len(ids)] = ids
        if i == 1:
            labels[i, 0] = 1 + (labels[i, 0] % 40)   # one substitution
    import ppasr_amd.parallel as par
    old = par.greedy_ids_decoder
    par.greedy_ids_decoder = lambda blank=0: _cpu_greedy
    try:
        val = ev.evaluate(model, [(x, labels, np.array(lens), None)], vocab, shard="buckets")
    finally:
        par.greedy_ids_decoder = old
    assert 0.0 < val < 0.02
