import sys, os, ctypes, numpy as np, torch, tempfile
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from lm_util import read_arpa, write_synthetic_arpa
from test_ctc_beam_gpu import _oracle, _probs
from test_ctc_beam_lm_gpu import _oracle_lm_decode, _vocab
from ppasr_amd.decoders.beam_search_decoder import Scorer, beam_search_ids
lib = _oracle()
bad = 0; total = 0
tmp = tempfile.mkdtemp()
for seed in range(60):
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    T = int(rng.integers(1, 120)); V = int(rng.choice([40, 97, 300, 1000, 4233])); beam = int(rng.choice([1, 2, 5, 10, 33, 100, 300]))
    order = int(rng.integers(2, 6)); use_lm = bool(rng.integers(0, 2)); kind = str(rng.choice(["peaky", "flat"]))
    cp = float(rng.choice([0.99, 0.9, 0.5])); tn = int(rng.choice([40, 10, 3]))
    alpha, beta = float(rng.uniform(0.2, 3)), float(rng.uniform(-2, 5))
    vocab = _vocab(V)
    lm = scorer = None
    if use_lm:
        known = [c for c in vocab[2:-1] if rng.random() < 0.8][:300]
        if len(known) < 3: continue
        arpa = write_synthetic_arpa(os.path.join(tmp, f"lm{seed}.arpa"), known, order=order, seed=seed)
        lm = read_arpa(arpa, vocab); scorer = Scorer(alpha, beta, arpa, vocab)
    p = _probs(rng, T, V, kind)
    nb = min(beam, 3)
    tk, ln, sc, _ = beam_search_ids(torch.from_numpy(p)[None].cuda(), beam, cp, tn, 0, nbest=nb, ext_scorer=scorer)
    ref = _oracle_lm_decode(lib, [p], V, beam, cp, tn, lm, alpha, beta, nb)
    got = tk[0, 0, :int(ln[0, 0])].cpu().numpy().tolist()
    total += 1
    if got != ref[0][0] or abs(float(sc[0, 0]) - ref[0][1]) > 2e-4 * max(1, abs(ref[0][1])):
        bad += 1
        print("MISMATCH", seed, T, V, beam, order, use_lm, kind, cp, tn, got[:8], ref[0][0][:8], float(sc[0,0]), ref[0][1])
print("fuzz done", total, "cases", bad, "mismatches")
