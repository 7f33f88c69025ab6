"""CPU model of the beam kernel's two selections (ctc_beam.hip (e) general / (e') staircase) on random tables: how often the
staircase verification passes, and that a passing frame gives the general selection's survivors."""
import math, sys
import numpy as np

NEG = -3.4028234663852886e38
def lse(a, b):
    if a <= NEG: return b
    if b <= NEG: return a
    m = max(a, b)
    return np.float32(math.log(math.exp(a - m) + math.exp(b - m)) + m)

def frame_lists(p, cutoff_prob, top_n):
    V = len(p)
    idx = np.arange(V)
    if cutoff_prob < 1.0 or top_n < V:
        order = np.lexsort((idx, -p))
        if cutoff_prob < 1.0:
            cum = np.cumsum(p[order].astype(np.float64))
            n = int(np.searchsorted(cum, cutoff_prob) + 1)
        else:
            n = V
        n = min(n, top_n, V)
        order = order[:n]
    else:
        order = idx
    return order, np.log(p[order] + np.float32(1.1754944e-38)).astype(np.float32)

def run(T, V, beam, cutoff_prob, top_n, kind, seed, margin=2):
    rng = np.random.default_rng(seed)
    logits = rng.standard_normal((T, V)).astype(np.float32) * (3 if kind == "flat3" else 1)
    if kind == "peaky":
        idx = np.repeat(rng.integers(0, V, size=(T + 2) // 3), 3)[:T]
        logits[np.arange(T), idx] += 6.0
        logits[:, 0] += np.where(rng.random(T) < 0.4, 7.0, 0.0).astype(np.float32)
    e = np.exp(logits - logits.max(1, keepdims=True)); P = (e / e.sum(1, keepdims=True)).astype(np.float32)
    # hypothesis = dict(node, chr, par, b, nb, score)
    hyps = [dict(node=0, chr=-1, par=-1, b=np.float32(0), nb=np.float32(NEG), score=np.float32(0))]
    n_nodes = 1
    att = ok = mism = 0
    for t in range(T):
        cand, lp = frame_lists(P[t], cutoff_prob, top_n)
        C, nb = len(cand), len(hyps)
        kidx = {int(c): k for k, c in enumerate(cand)}
        exists = set()
        new = []
        for q, h in enumerate(hyps):
            bc = lp[kidx[0]] + h["score"] if 0 in kidx else np.float32(NEG)
            nbc = np.float32(NEG)
            cq = h["chr"]
            if cq in kidx and cq != 0:
                nbc = lp[kidx[cq]] + h["nb"]
                pi = [i for i, g in enumerate(hyps) if g["node"] == h["par"]]
                if pi:
                    pi = pi[-1]; g = hyps[pi]
                    lpx = np.float32(NEG)
                    if cq == g["chr"]:
                        if g["b"] > NEG: lpx = lp[kidx[cq]] + g["b"]
                    else: lpx = lp[kidx[cq]] + g["score"]
                    nbc = lse(nbc, lpx)
                    exists.add((pi, kidx[cq]))
            new.append((np.float32(bc), np.float32(nbc), lse(bc, nbc)))
        # general: all elements
        elems = []  # (score, char, e, kind, i, k)
        for q, h in enumerate(hyps): elems.append((new[q][2], h["chr"], q, "stay", q, -1))
        for i, h in enumerate(hyps):
            for k in range(C):
                c = int(cand[k])
                if c == 0 or (i, k) in exists: continue
                if c == h["chr"]: v = lp[k] + h["b"] if h["b"] > NEG else np.float32(NEG)
                else: v = lp[k] + h["score"]
                elems.append((np.float32(v), c, nb + i * C + k, "child", i, k))
        keyf = lambda x: (-float(x[0]), x[1] + 1, x[2])
        n_valid = len(elems)
        k_sel = min(beam, n_valid)
        gen = sorted(sorted(elems, key=keyf)[:k_sel], key=lambda x: x[2])
        # staircase
        if n_valid > beam and C > 0:
            rank = sorted(range(nb), key=lambda q: (-float(hyps[q]["score"]), q))
            Ks = [min(C, beam // (r + 1) + margin) for r in range(nb)]
            if nb + sum(Ks) <= 128:
                att += 1
                S = [x for x in elems if x[3] == "stay"]
                allow = {(rank[r], k) for r in range(nb) for k in range(Ks[r])}
                S += [x for x in elems if x[3] == "child" and (x[4], x[5]) in allow]
                top = sorted(S, key=keyf)[:beam]
                good = len(top) == beam
                if good:
                    thr = float(top[-1][0])
                    for r in range(nb):
                        if Ks[r] < C:
                            ub = np.float32(lp[Ks[r]] + hyps[rank[r]]["score"])
                            if not (float(ub) < thr): good = False
                if good:
                    ok += 1
                    if sorted(top, key=lambda x: x[2]) != gen: mism += 1
        nxt = []
        for pos, x in enumerate(gen):
            if x[3] == "stay":
                h = hyps[x[4]]
                nxt.append(dict(node=h["node"], chr=h["chr"], par=h["par"], b=new[x[4]][0], nb=new[x[4]][1], score=new[x[4]][2]))
            else:
                nxt.append(dict(node=n_nodes + pos, chr=x[1], par=hyps[x[4]]["node"], b=np.float32(NEG), nb=x[0], score=x[0]))
        n_nodes += k_sel
        hyps = nxt
    return att, ok, mism

for cfg in [(120, 4233, 10, 0.99, 40, "flat3"), (120, 500, 10, 0.99, 40, "peaky"), (60, 700, 13, 0.9, 7, "flat"),
            (60, 700, 13, 0.9, 7, "peaky"), (100, 90, 16, 0.999, 40, "flat"), (50, 30, 8, 1.0, 40, "flat")]:
    for margin in (2, 4):
        print(cfg, "margin", margin, "-> attempted, passed, mismatching:", run(*cfg, seed=1, margin=margin))
