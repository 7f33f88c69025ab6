"""GPU parity of the HIP fbank front-end (csrc/fbank.hip) vs oracle/fbank_oracle.py (float64 restatement of
AudioSegment.normalize -> int16 -> Kaldi fbank).  Parity unpinned: paddleaudio is not importable offline."""
import numpy as np
import pytest
import torch

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


def test_mean_square_bit_exact_for_every_last_chunk_length():
    """The mean square is summed in numpy's order (pairwise tree per 8192-sample chunk).  A SHORT last chunk's tree can be
    one level deeper than a full chunk's (8191 -> ... -> 135 -> 71: depth 7); lengths with n % 8192 in 7689..8191 used to
    read unwritten LDS (round-5 advisor finding).  Sweep that whole range, plus a coarse sweep of the rest, with zero, one
    and two preceding full chunks: the per-chunk sums the kernel leaves in its workspace, added in chunk order, must equal
    np.add.reduce(x ** 2) bit for bit, and the gain must be db_gain's to a few ulp (numpy's float32 log10 is libm's
    log10f, the kernel rounds a double log10: one ulp of rms_db is two of the gain)."""
    from ppasr_amd.data_utils.featurizer import AudioFeaturizer, db_gain
    f = AudioFeaturizer(n_mels=80, sample_rate=16000, use_dB_normalization=True, target_dB=-20)
    base = _audio(1.6, seed=11)
    assert base.size >= 3 * 8192
    tails = list(range(7600, 8193)) + list(range(400, 7600, 97)) + [128, 129, 135, 136, 143, 144, 263, 519, 1031, 2055, 4103]
    bad, bad_gain = [], []
    for full in (0, 1, 2):
        for tail in tails:
            n = full * 8192 + tail
            if n < 400:
                continue
            wav = base[:n]
            f.featurize_device(wav)
            chunks = (n + 8191) // 8192
            sums = f._ws[:4 * chunks].view(torch.float32).cpu().numpy()
            acc = np.float32(0)
            for v in sums:
                acc = np.float32(acc + v)
            if acc.tobytes() != np.add.reduce(wav ** 2).tobytes():
                bad.append(n)
            g = np.float32(db_gain(wav, -20))
            if abs(float(np.float32(f.last_gain)) - float(g)) > 4 * float(np.spacing(g)):
                bad_gain.append(n)
    assert not bad, bad[:20]
    assert not bad_gain, bad_gain[:20]


def test_normalize_refuses_gain_beyond_max_gain_db():
    """AudioSegment.normalize raises ValueError when the gain exceeds max_gain_db = 300 (data_utils/audio.py:301-303)."""
    from ppasr_amd.data_utils.featurizer import AudioFeaturizer, db_gain
    wav = np.full(1600, 1e-20, np.float32)  # mean square 1e-40 (denormal, non-zero): rms -400 dB, gain 380 dB
    with pytest.raises(ValueError):
        db_gain(wav, -20)
    with pytest.raises(ValueError):
        AudioFeaturizer(n_mels=80, sample_rate=16000).featurize(wav)
    assert db_gain(np.zeros(1600, np.float32), -20) == np.float32(0.1)  # all-zero: rms_db 0 by definition (audio.py:527)
