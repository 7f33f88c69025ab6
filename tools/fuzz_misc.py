"""Ad-hoc sweep of the smaller entry points: fbank (random lengths / sample rates / gains), greedy decode edge tables,
DeepSpeech2 (random batch lens, random chunk lengths with state carry).  Compares with the oracles; prints a summary."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ctc_decoders_oracle as dec  # noqa: E402
from oracle import fbank_oracle  # noqa: E402
from oracle.deepspeech2_oracle import DeepSpeech2Oracle  # noqa: E402
from ppasr_amd.utils.synth import deepspeech2_state_dict, synth_features  # noqa: E402

bad = 0
n = 0


def check(ok, *what):
    global bad, n
    n += 1
    if not ok:
        bad += 1
        print("FAIL", *what)


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# ---------------- fbank ----------------
from ppasr_amd.data_utils.featurizer import AudioFeaturizer  # noqa: E402

rng = np.random.Generator(np.random.PCG64(4242))
for case in range(40):
    sr = int(rng.choice([8000, 16000]))
    mel = int(rng.choice([40, 80]))
    win = sr * 25 // 1000
    ns = int(rng.choice([win - 1, win, win + 1, win + sr // 100 - 1, win + sr // 100, int(rng.integers(win, 20 * sr))]))
    amp = float(10 ** rng.uniform(-4, 0))
    t = np.arange(ns) / sr
    x = amp * (np.sin(2 * np.pi * rng.uniform(60, sr / 2.2) * t) + 0.3 * rng.standard_normal(ns))
    x = np.clip(x, -1, 1).astype(np.float32)
    use_db = bool(rng.integers(0, 2))
    tgt = float(rng.choice([-20.0, -26.0, -10.0]))
    f = AudioFeaturizer(feature_method="fbank", n_mels=mel, sample_rate=sr, use_dB_normalization=use_db, target_dB=tgt)
    try:
        got = f.featurize(x, sr)
        ref = fbank_oracle.featurize(x, sr, mel, use_db, tgt)
        ok = got.shape == ref.shape and (got.size == 0 or float(np.abs(got - ref).max()) < 2e-3)
        check(ok, "fbank", sr, mel, ns, amp, use_db, tgt, got.shape, ref.shape,
              None if got.size == 0 or got.shape != ref.shape else float(np.abs(got - ref).max()))
    except Exception as e:  # noqa: BLE001
        check(False, "fbank ERROR", sr, mel, ns, repr(e)[:160])
print("fbank cases done", n, "bad", bad)

# ---------------- greedy ----------------
from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decode_ids  # noqa: E402

for case in range(30):
    B = int(rng.integers(1, 40))
    T = int(rng.integers(1, 900))
    V = int(rng.choice([2, 3, 29, 100, 4233, 5000]))
    kind = case % 5
    p = rng.random((B, T, V)).astype(np.float32)
    if kind == 0:
        p[:, :, 0] = 2.0                     # all blank
    elif kind == 1:
        p[:, :, V - 1] = 2.0                 # one token repeated for ever -> a single output
    elif kind == 2:
        p[:] = 0.25                          # every entry ties -> index 0 = blank
    elif kind == 3:
        p[:, ::2, 0] = 2.0                   # blank / token alternation: every token survives
        p[:, 1::2, 1 % V] = 3.0
    lens = rng.integers(0, T + 1, size=B).astype(np.int32)
    if kind == 4:
        lens[:] = 0
    tokens, cnt, score, fa, fp = greedy_decode_ids(torch.from_numpy(p).cuda(), lens)
    torch.cuda.synchronize()
    ok = True
    for b in range(B):
        nb = int(lens[b])
        ids, mi, mp = dec.greedy_tokens(p[b, :nb])
        ok &= np.array_equal(tokens[b, :int(cnt[b])].cpu().numpy(), ids) and int(cnt[b]) == len(ids)
        ok &= np.array_equal(fa[b, :nb].cpu().numpy(), mi)
        refs = float(np.mean(mp.astype(np.float64))) * 100 if len(mp) else 0.0
        ok &= abs(float(score[b]) - refs) <= 1e-9 * max(1.0, refs)
    check(ok, "greedy", B, T, V, kind)
print("greedy cases done", n, "bad", bad)

# ---------------- DeepSpeech2 ----------------
from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model  # noqa: E402

for streaming in (True, False):
    V, L = 150, 3
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, streaming=streaming, seed=77 + streaming, perturb_norm=True)
    model = DeepSpeech2Model(80, V, streaming=streaming, encoder_conf=dict(num_rnn_layers=L, rnn_size=1024), state_dict=sd,
                             device="cuda:0")
    oracle = DeepSpeech2Oracle(sd, L, 1024, streaming)
    for case in range(8):
        B = int(rng.integers(1, 6))
        T = int(rng.integers(9, 260))
        lens = np.sort(rng.integers(9, T + 1, size=B))[::-1].copy()
        lens[0] = T
        x, _ = synth_features(B, T, lens=lens.tolist(), seed=1000 + case)
        try:
            probs, ol, fh, fc = model.get_encoder_out_chunk(x, lens)
            torch.cuda.synchronize()
            rp, rl, rh, rc = oracle.forward(x, lens)
            ok = ol.cpu().tolist() == rl.tolist() and rel(probs.cpu().numpy(), rp.numpy()) < 1e-3
            ok &= rel(fh.cpu().numpy(), rh.numpy()) < 1e-3 and rel(fc.cpu().numpy(), rc.numpy()) < 1e-3
            check(ok, "ds2 batch", streaming, B, T, lens.tolist())
        except Exception as e:  # noqa: BLE001
            check(False, "ds2 ERROR", streaming, B, T, lens.tolist(), repr(e)[:160])
    if streaming:
        # state carry over random chunk lengths
        x, _ = synth_features(1, 600, seed=31)
        h = c = rh = rc = None
        s = 0
        while s < 600 - 9:
            ln = int(rng.integers(9, 120))
            chunk = x[:, s:s + ln]
            ll = np.array([chunk.shape[1]])
            probs, _, h, c = model.get_encoder_out_chunk(chunk, ll, h, c)
            rp, _, rh, rc = oracle.forward(chunk, ll, rh, rc)
            check(rel(probs.cpu().numpy(), rp.numpy()) < 1e-3, "ds2 chunk", s, ln, rel(probs.cpu().numpy(), rp.numpy()))
            s += ln
print("fuzz_misc done:", n, "cases,", bad, "problems")
