"""Where does the HIP fbank deviate most from the float64 oracle?  (diagnostic, GPU)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import fbank_oracle
from tests.test_fbank_gpu import _audio
from ppasr_amd.data_utils.featurizer import AudioFeaturizer

for seconds in (10.0, 29.97):
    wav = _audio(seconds, seed=int(seconds * 10))
    f = AudioFeaturizer(feature_method="fbank", n_mels=80, sample_rate=16000, use_dB_normalization=True, target_dB=-20)
    got = f.featurize(wav).astype(np.float64)
    ref = fbank_oracle.featurize(wav, 16000, 80, True, -20.0)
    err = np.abs(got - ref)
    fr, mb = np.unravel_index(err.argmax(), err.shape)
    print(f"{seconds}s: max err {err.max():.3e} at frame {fr} mel {mb}: got {got[fr, mb]:.6f} ref {ref[fr, mb]:.6f}; "
          f"entries > 1e-4: {(err > 1e-4).sum()} of {err.size}; frames with any > 1e-4: {(err > 1e-4).any(1).sum()}")
    print("   row err:", np.array2string(err[fr], precision=1, max_line_width=200))
    print("   row ref:", np.array2string(ref[fr], precision=2, max_line_width=200))
    worst = np.argsort(err.max(1))[-5:]
    print("   worst frames:", worst, err.max(1)[worst])
    i16 = fbank_oracle.normalize_to_int16(wav, True, -20.0)
    seg = i16[fr * 160: fr * 160 + 400]
    print("   int16 frame: min", seg.min(), "max", seg.max(), "clipped", int((np.abs(seg) >= 32767).sum()))
    for c in (4e-6, 1e-5, 2e-5):
        tol = 2e-4 + c * np.exp(0.5 * (ref.max(axis=1, keepdims=True) - ref))
        r = err / tol
        k = np.unravel_index(r.argmax(), r.shape)
        print(f"   c={c:g}: max err/tol {r.max():.2f} at {k}: err {err[k]:.2e} ref {ref[k]:.2f} frame max {ref[k[0]].max():.2f}")
