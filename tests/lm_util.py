"""Test helpers for the external scorer: a synthetic character-based ARPA model and an (independent) ARPA reader that
feeds the C oracle."""
import numpy as np

LM_MAX_ORDER = 6


def write_synthetic_arpa(path, chars, order=3, n_sent=120, sent_len=12, seed=0):
    """Random back-off n-gram model over `chars` (single-character words): n-grams collected from random sentences
    (so every n-gram's prefix exists), log10 probabilities / back-offs drawn at random (need not normalise)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    grams = [set() for _ in range(order)]
    # favour a few characters so that higher orders repeat
    weights = rng.random(len(chars)) ** 3
    weights /= weights.sum()
    for _ in range(n_sent):
        n = int(rng.integers(2, sent_len))
        s = ["<s>"] + [chars[i] for i in rng.choice(len(chars), size=n, p=weights)] + ["</s>"]
        for k in range(1, order + 1):
            for i in range(len(s) - k + 1):
                g = tuple(s[i:i + k])
                if k > 1 and g[-1] == "<s>":
                    continue
                grams[k - 1].add(g)
    for c in chars:
        grams[0].add((c,))
    grams[0].update({("<unk>",), ("<s>",), ("</s>",)})
    with open(path, "w", encoding="utf-8") as f:
        f.write("\\data\\\n")
        for k in range(order):
            f.write(f"ngram {k + 1}={len(grams[k])}\n")
        for k in range(order):
            f.write(f"\n\\{k + 1}-grams:\n")
            for g in sorted(grams[k]):
                p = -99.0 if g == ("<s>",) else -float(rng.random() * 3.5 + 0.05)
                line = f"{p:.6f}\t{' '.join(g)}"
                if k + 1 < order and g[-1] != "</s>":
                    line += f"\t{-float(rng.random() * 1.2):.6f}"
                f.write(line + "\n")
        f.write("\n\\end\\\n")
    return path


def read_arpa(path, vocabulary):
    """-> dict(order, gram_n [N], gram_w [N,6], prob [N], backoff [N], tok2lm [V], bos, eos); word ids: <unk> = 0."""
    words = {"<unk>": 0}
    rows = []
    section = 0
    with open(path, encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith("\\"):
                if line.endswith("-grams:"):
                    section = int(line[1:line.index("-")])
                elif line == "\\end\\":
                    break
                else:
                    section = 0
                continue
            if section == 0:
                continue
            parts = line.split()
            ws = parts[1:1 + section]
            for w in ws:
                words.setdefault(w, len(words))
            bo = float(parts[1 + section]) if len(parts) > 1 + section else 0.0
            rows.append((section, [words[w] for w in ws], float(parts[0]), bo))
    N = len(rows)
    gram_n = np.zeros(N, np.int32)
    gram_w = np.zeros((N, LM_MAX_ORDER), np.int32)
    prob = np.zeros(N, np.float32)
    backoff = np.zeros(N, np.float32)
    for i, (n, ws, p, b) in enumerate(rows):
        gram_n[i] = n
        gram_w[i, :n] = ws
        prob[i] = np.float32(p)
        backoff[i] = np.float32(b)
    tok2lm = np.array([0 if t == " " else words.get(t, 0) for t in vocabulary], np.int32)
    return dict(order=int(gram_n.max()), gram_n=gram_n, gram_w=gram_w, prob=prob, backoff=backoff, tok2lm=tok2lm,
                bos=words["<s>"], eos=words["</s>"], words=dict(words))
