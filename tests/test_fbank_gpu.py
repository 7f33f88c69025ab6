"""GPU parity of the HIP fbank front-end (csrc/fbank.hip) vs oracle/fbank_oracle.py (float64 restatement of
AudioSegment.normalize -> int16 -> Kaldi fbank).  Parity unpinned: paddleaudio is not importable offline."""
import numpy as np
import pytest

from oracle import fbank_oracle

pytestmark = pytest.mark.gpu


def _audio(seconds, seed=0, sr=16000):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(int(sr * seconds)) / sr
    x = 0.2 * np.sin(2 * np.pi * 180 * t) * (1 + 0.5 * np.sin(2 * np.pi * 2.5 * t)) + 0.05 * rng.standard_normal(t.shape)
    x += 0.1 * np.sin(2 * np.pi * (500 + 800 * t) * t)
    return x.astype(np.float32)


@pytest.mark.parametrize("seconds,use_db", [(2.0, True), (2.0, False), (0.0251, True), (10.0, True), (0.5123, True), (1.0240625, True),
                                            (29.97, True)])
def test_fbank_matches_oracle(seconds, use_db):
    from ppasr_amd.data_utils.featurizer import AudioFeaturizer
    wav = _audio(seconds, seed=int(seconds * 10))
    f = AudioFeaturizer(feature_method="fbank", n_mels=80, sample_rate=16000, use_dB_normalization=use_db, target_dB=-20)
    got = f.featurize(wav)
    ref = fbank_oracle.featurize(wav, 16000, 80, use_db, -20.0)
    assert got.shape == ref.shape and got.dtype == np.float32
    assert got.shape[0] == 1 + (len(wav) - 400) // 160
    # The gain is bit-exact (the kernel sums the squares in numpy's order: pairwise, 8192-element chunks), so every int16
    # sample equals the oracle's; what is left is fp32 arithmetic (window, radix-2 FFT, mel sums) against the oracle's
    # float64.  Its error scales with the frame's LARGEST spectral amplitude, so a mel bin that sits e_max / e below the
    # frame's peak carries a relative energy error ~ sqrt(e_max / e): measured <= 2.2e-5 * sqrt(e_max / e) (tools/
    # experiments/r05/fbank_diag.py) -- below 2e-4 for most bins, 1.2e-2 in a spectral null 21 nepers below a sine (the
    # 29.97 s case, frame 1716; paddleaudio's own fbank is an fp32 rfft with the same property).
    err = np.abs(got.astype(np.float64) - ref)
    tol = 2e-4 + 4e-5 * np.exp(0.5 * (ref.max(axis=1, keepdims=True) - ref))
    print("max abs err", float(err.max()), "max err / tol", float((err / tol).max()))
    assert (err <= tol).all()
    assert np.median(err) < 2e-5


def test_fbank_edge_cases():
    from ppasr_amd.data_utils.featurizer import AudioFeaturizer
    f = AudioFeaturizer(n_mels=80, sample_rate=16000)
    assert f.featurize(np.zeros(399, np.float32)).shape == (0, 80)      # shorter than one window
    z = f.featurize(np.zeros(16000, np.float32))                         # silence: log(eps) floor everywhere
    assert z.shape == (98, 80) and np.allclose(z, np.log(np.finfo(np.float32).eps))
    big = f.featurize(np.full(1600, 0.999, np.float32))                  # DC only -> removed -> floor
    assert np.allclose(big, np.log(np.finfo(np.float32).eps), atol=1e-3)
    f40 = AudioFeaturizer(n_mels=40, sample_rate=8000)                   # 8 kHz: 200-sample window, 256-point FFT
    w = _audio(1.0, seed=3, sr=8000)
    ref = fbank_oracle.featurize(w, 8000, 40, True, -20.0)
    assert np.abs(f40.featurize(w, 8000) - ref).max() < 1e-3
