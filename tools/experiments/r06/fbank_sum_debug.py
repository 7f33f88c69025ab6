"""debug: per-chunk sums of squares left in the fbank workspace vs numpy, for a few lengths"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
from ppasr_amd.data_utils.featurizer import AudioFeaturizer, db_gain
from test_fbank_gpu import _audio
f = AudioFeaturizer(n_mels=80, sample_rate=16000, use_dB_normalization=True, target_dB=-20)
base = _audio(1.6, seed=11)
print(np.__version__, np.show_config is not None)
for n in (400, 500, 1000, 4096, 7600, 7689, 8191, 8192, 8192 + 500, 2 * 8192 + 8000):
    wav = base[:n]
    f.featurize_device(wav)
    chunks = (n + 8191) // 8192
    sums = f._ws[:4 * chunks].view(torch.float32).cpu().numpy()
    want = [np.add.reduce(wav[c * 8192:(c + 1) * 8192] ** 2) for c in range(chunks)]
    tot = np.add.reduce(wav ** 2)
    acc = np.float32(0)
    for v in sums: acc = np.float32(acc + v)
    acc2 = np.float32(0)
    for v in want: acc2 = np.float32(acc2 + v)
    print(n, [float(s).hex() for s in sums], [float(w).hex() for w in want], float(tot).hex(), float(acc).hex(), float(acc2).hex(),
          "gain", float(f.last_gain).hex(), float(db_gain(wav, -20)).hex())
