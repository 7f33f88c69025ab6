"""Host-side feature front-end that feeds the hot path (NOT part of the hand-written-kernel scope:
SURVEY.md §8f ranks a HIP fbank as the next row).  Mirrors the inference behaviour of
``ppasr/data_utils/featurizer/audio_featurizer.py`` (dB normalisation -> int16 scale -> Kaldi fbank,
dither 0 at inference, :37-67,120-138) and ``text_featurizer.py`` (vocabulary file loader).

The fbank follows Kaldi's published algorithm with the defaults of
``paddleaudio.compliance.kaldi.fbank`` (snip_edges, remove DC, pre-emphasis 0.97, povey window,
512-point FFT, power spectrum, 80 mel bins from 20 Hz to Nyquist, log with float eps floor).
paddleaudio is not installable offline, so this front-end is "parity unpinned".
"""
import io
import math
import wave

import numpy as np
import torch

__all__ = ["AudioFeaturizer", "TextFeaturizer", "load_audio"]

_EPS = float(np.finfo(np.float32).eps)


def load_audio(audio_data, sample_rate=16000):
    """-> (float32 samples in [-1,1] mono, sample_rate).  Accepts a .wav path, a binary file object,
    wav-file bytes or a numpy array (predict.py:143-160; only PCM wav is supported without soundfile)."""
    if isinstance(audio_data, np.ndarray):
        x = audio_data
        if x.dtype.kind == "i":
            x = x.astype(np.float32) / float(2 ** (8 * x.dtype.itemsize - 1))
        return x.astype(np.float32).reshape(-1) if x.ndim == 1 else x.astype(np.float32).mean(axis=1), sample_rate
    if isinstance(audio_data, (bytes, bytearray)):
        audio_data = io.BytesIO(audio_data)
    if isinstance(audio_data, str) or hasattr(audio_data, "read"):
        with wave.open(audio_data, "rb") as w:
            sr, ch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
            raw = w.readframes(n)
        return pcm_bytes_to_float(raw, ch, sw), sr
    raise Exception(f"unsupported audio_data type: {type(audio_data)}")


def pcm_bytes_to_float(data, channels=1, samp_width=2):
    """AudioSegment.from_pcm_bytes (data_utils/audio.py:122-139)."""
    dt = {1: np.int8, 2: np.int16, 4: np.int32}[samp_width]
    x = np.frombuffer(data, dtype=dt).astype(np.float32) / float(2 ** (8 * samp_width - 1))
    if channels > 1:
        x = x.reshape(-1, channels).mean(axis=1)
    return x


def _mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def _mel_banks(n_mels, n_fft, sr, low=20.0, high=0.0):
    nyq = 0.5 * sr
    high = high + nyq if high <= 0 else high
    fft_bin_width = sr / n_fft
    mel_lo, mel_hi = _mel(low), _mel(high)
    delta = (mel_hi - mel_lo) / (n_mels + 1)
    b = np.arange(n_mels)[:, None]
    left, center, right = mel_lo + b * delta, mel_lo + (b + 1) * delta, mel_lo + (b + 2) * delta
    mel = _mel(fft_bin_width * np.arange(n_fft // 2))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return np.maximum(0.0, np.minimum(up, down)).astype(np.float32)  # [n_mels, n_fft/2]


class AudioFeaturizer:
    def __init__(self, feature_method="fbank", n_mels=80, n_mfcc=40, sample_rate=16000, use_dB_normalization=True,
                 target_dB=-20, train=False, device=None, **_ignored):
        if feature_method != "fbank":
            raise NotImplementedError("only feature_method='fbank' is on the hot path")
        self._n_mels = n_mels
        self._sr = sample_rate
        self._use_db = use_dB_normalization
        self._target_db = target_dB
        self._device = device
        self._win = int(sample_rate * 0.025)
        self._shift = int(sample_rate * 0.010)
        self._nfft = 1 << (self._win - 1).bit_length()
        self._banks = None

    @property
    def feature_dim(self):
        return self._n_mels

    def featurize(self, samples, sample_rate=None):
        """float32 mono samples -> fbank [T, n_mels] float32 (numpy)."""
        x = np.asarray(samples, np.float32).copy()
        sr = sample_rate or self._sr
        if sr != self._sr:
            raise NotImplementedError("resampling is outside the hot path; feed audio at the model's sample rate")
        if self._use_db:  # AudioSegment.normalize (audio.py:287-304)
            ms = float(np.mean(x ** 2)) if x.size else 0.0
            rms_db = 10 * math.log10(ms if ms != 0 else 1)
            x *= 10.0 ** ((self._target_db - rms_db) / 20.0)
        x = np.clip(x * 32768.0, -32768, 32767).astype(np.int16).astype(np.float32)  # .to('int16') (audio.py:244)
        if len(x) < self._win:
            return np.zeros((0, self._n_mels), np.float32)
        dev = self._device or ("cuda" if torch.cuda.is_available() else "cpu")
        w = torch.from_numpy(x).to(dev)
        n = 1 + (len(x) - self._win) // self._shift  # snip_edges
        frames = w.unfold(0, self._win, self._shift)[:n]
        frames = frames - frames.mean(dim=1, keepdim=True)  # remove_dc_offset
        prev = torch.cat([frames[:, :1], frames[:, :-1]], dim=1)
        frames = frames - 0.97 * prev  # pre-emphasis
        win = torch.hann_window(self._win, periodic=False, dtype=torch.float32, device=dev).pow(0.85)  # povey
        frames = frames * win
        spec = torch.fft.rfft(frames, n=self._nfft).abs().pow(2.0)  # power spectrum [n, nfft/2+1]
        if self._banks is None or self._banks.device != spec.device:
            self._banks = torch.from_numpy(_mel_banks(self._n_mels, self._nfft, self._sr)).to(spec.device)
        mel = spec[:, : self._nfft // 2] @ self._banks.T
        return torch.log(torch.clamp(mel, min=_EPS)).cpu().numpy().astype(np.float32)


class TextFeaturizer:
    """Vocabulary loader (text_featurizer.py:8-59): one token per line, optional tab-separated count."""

    def __init__(self, vocab_filepath=None, vocab_list=None):
        if vocab_list is None:
            with open(vocab_filepath, "r", encoding="utf-8") as f:
                vocab_list = [line.split("\t")[0].replace("\n", "") for line in f.readlines()]
        self._vocab_list = list(vocab_list)
        self._vocab_dict = {t: i for i, t in enumerate(self._vocab_list)}
        self.unk = "<unk>"

    @property
    def vocab_size(self):
        return len(self._vocab_list)

    @property
    def vocab_list(self):
        return self._vocab_list

    def featurize(self, text):
        ids = []
        for tok in list(text.strip()):
            tok = "<space>" if tok == " " else tok
            ids.append(self._vocab_dict.get(tok, self._vocab_dict.get(self.unk, 0)))
        return ids
