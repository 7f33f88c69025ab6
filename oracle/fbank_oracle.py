"""ORACLE (test infrastructure, not product code) -- numpy float64 restatement of the feature front-end that
precedes the hot path: ``AudioFeaturizer.featurize`` (ppasr/data_utils/featurizer/audio_featurizer.py:37-67,120-138)
= ``AudioSegment.normalize`` (data_utils/audio.py:287-304) -> ``.to('int16')`` (audio.py:244) ->
``paddleaudio.compliance.kaldi.fbank`` (third-party dependency ``paddleaudio>=1.0.1``, requirements.txt:14, NOT in
/root/reference and not installable offline).  The fbank follows Kaldi's published algorithm with that function's
defaults (snip_edges, remove_dc_offset, pre-emphasis 0.97, povey window, round-to-power-of-two FFT, power spectrum,
triangular mel banks 20 Hz .. Nyquist, log with FLT_EPSILON floor, dither 0 at inference).  PARITY UNPINNED."""
import math

import numpy as np

EPS = float(np.finfo(np.float32).eps)


def normalize_to_int16(samples, use_db_normalization=True, target_db=-20.0):
    x = np.asarray(samples, np.float32).copy()
    if use_db_normalization:
        # AudioSegment.rms_db / normalize / gain_db (data_utils/audio.py:519-530,287-304,256-264) with the scalar types numpy
        # 1.x gives them there (the reference needs numpy 1.x: it reads np.sctypes): the mean square is float32 (np.mean of
        # the float32 squares: numpy's pairwise sums over 8192-element chunks), its log10 is float32; `10 * <float32
        # scalar>`, `target_db - rms_db` and `gain / 20.` are float64 (legacy promotion of all-scalar operands), the power
        # is taken in float64 and rounded to float32 when it scales the float32 samples.  Pinned by tests/golden/
        # ref_wav.npz (the reference's source on its own dataset/test.wav, make_wav_goldens.py emulating numpy 1.x).
        ms = np.mean(x ** 2) if x.size else np.float32(0.0)
        rms_db = 10.0 * float(np.log10(ms)) if ms != 0 else 0.0
        x *= np.float32(10.0 ** ((float(target_db) - rms_db) / 20.0))
    return np.clip(x * np.float32(32768.0), -32768, 32767).astype(np.int16)


def mel_banks(n_mels, n_fft, sr, low=20.0):
    mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, np.float64) / 700.0)
    mel_lo, mel_hi = mel(low), mel(0.5 * sr)
    delta = (mel_hi - mel_lo) / (n_mels + 1)
    b = np.arange(n_mels)[:, None]
    left, center, right = mel_lo + b * delta, mel_lo + (b + 1) * delta, mel_lo + (b + 2) * delta
    m = mel(sr / n_fft * np.arange(n_fft // 2))[None, :]
    return np.maximum(0.0, np.minimum((m - left) / (center - left), (right - m) / (right - center)))


def kaldi_fbank(int16_samples, sr=16000, n_mels=80, frame_length_ms=25.0, frame_shift_ms=10.0):
    x = np.asarray(int16_samples).astype(np.float64)
    win, shift = int(sr * 0.001 * frame_length_ms), int(sr * 0.001 * frame_shift_ms)
    if len(x) < win:
        return np.zeros((0, n_mels), np.float64)
    n = 1 + (len(x) - win) // shift
    nfft = 1 << (win - 1).bit_length()
    idx = np.arange(win)[None, :] + shift * np.arange(n)[:, None]
    fr = x[idx]
    fr = fr - fr.mean(axis=1, keepdims=True)
    fr = fr - 0.97 * np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)
    fr = fr * np.power(0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win) / (win - 1)), 0.85)
    spec = np.abs(np.fft.rfft(fr, n=nfft, axis=1)) ** 2
    e = spec[:, : nfft // 2] @ mel_banks(n_mels, nfft, sr).T
    return np.log(np.maximum(e, EPS))


def featurize(samples, sr=16000, n_mels=80, use_db_normalization=True, target_db=-20.0):
    return kaldi_fbank(normalize_to_int16(samples, use_db_normalization, target_db), sr, n_mels)
