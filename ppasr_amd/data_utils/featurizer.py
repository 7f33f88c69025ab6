"""Feature front-end that feeds the hot path: the host-side mirror of
``ppasr/data_utils/featurizer/audio_featurizer.py`` (dB normalisation -> int16 scale -> Kaldi fbank, dither 0 at
inference, :37-67,120-138) and ``text_featurizer.py`` (vocabulary file loader).

The arithmetic runs in ``csrc/fbank.hip`` behind ``ppasr_fbank_*`` (one workgroup per frame, frame resident in LDS
from the raw samples to the 80 log-mel values); this module only loads audio and moves buffers.  There is no CPU
path: without a HIP device ``featurize`` raises.  paddleaudio is not installable offline, so the front-end is
"parity unpinned" (checked against ``oracle/fbank_oracle.py``, a float64 restatement of Kaldi's algorithm).
"""
import ctypes
import io
import wave

import numpy as np
import torch

__all__ = ["AudioFeaturizer", "TextFeaturizer", "load_audio", "db_gain"]



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


def resample(samples, sample_rate, target_sample_rate):
    """``AudioSegment.resample`` (data_utils/audio.py:306-317).  The reference calls ``resampy.resample(...,
    filter='kaiser_best')``; resampy is not installable offline, so this is scipy's polyphase resampler with a Kaiser
    window (``scipy.signal.resample_poly``): the same kind of band-limited interpolation, not bit-identical to resampy.
    Host code on the audio I/O side of the path."""
    if int(sample_rate) == int(target_sample_rate):
        return np.asarray(samples, np.float32)
    from math import gcd

    from scipy.signal import resample_poly
    g = gcd(int(sample_rate), int(target_sample_rate))
    up, down = int(target_sample_rate) // g, int(sample_rate) // g
    return resample_poly(np.asarray(samples, np.float64), up, down, window=("kaiser", 14.769656459379492)).astype(np.float32)


def db_gain(samples, target_db=-20.0):
    """Gain factor of ``AudioSegment.normalize(target_db)`` (data_utils/audio.py:287-304 -> rms_db :519-530 -> gain_db
    :256-264) with the scalar types numpy 1.x gives the reference there: the mean square and its log10 in float32, ``10 *
    log10``, ``target_db - rms_db`` and the power in float64, rounded to float32 when it scales the float32 samples (the
    arithmetic of csrc/fbank.hip, pinned by tests/golden/ref_wav.npz).  Host-side: ``predict_stream`` needs it because the
    reference normalises its buffered ``remained_wav`` IN PLACE (see ppasr_amd/predict.py)."""
    x = np.asarray(samples, np.float32)
    ms = np.mean(x ** 2) if x.size else np.float32(0.0)
    rms_db = 10.0 * float(np.log10(ms)) if ms != 0 else 0.0
    check_gain_db(float(target_db) - rms_db, target_db)
    return np.float32(10.0 ** ((float(target_db) - rms_db) / 20.0))


MAX_GAIN_DB = 300.0  # AudioSegment.normalize's max_gain_db default (data_utils/audio.py:287)


def check_gain_db(gain, target_db):
    """``AudioSegment.normalize`` refuses a gain beyond max_gain_db (audio.py:301-303): same exception type and text."""
    if gain > MAX_GAIN_DB:
        raise ValueError(f"无法将段规范化到{target_db}dB，音频增益{gain}增益已经超过max_gain_db ({MAX_GAIN_DB}dB)")


def pcm_bytes_to_float(data, channels=1, samp_width=2):
    """AudioSegment.from_pcm_bytes (data_utils/audio.py:122-139)."""
    dt = {1: np.int8, 2: np.int16, 4: np.int32}[samp_width]
    x = np.frombuffer(data, dtype=dt).astype(np.float32) / float(2 ** (8 * samp_width - 1))
    if channels > 1:
        x = x.reshape(-1, channels).mean(axis=1)
    return x


class AudioFeaturizer:
    def __init__(self, feature_method="fbank", n_mels=80, n_mfcc=40, sample_rate=16000, use_dB_normalization=True,
                 target_dB=-20, train=False, device=None, **_ignored):
        if feature_method != "fbank":
            raise NotImplementedError("only feature_method='fbank' is on the hot path")
        self._n_mels = n_mels
        self._sr = sample_rate
        self._use_db = use_dB_normalization
        self._target_db = target_dB
        self._device = torch.device(device or "cuda:0")
        self._h = None
        self._ws = None

    def _handle(self):
        if self._h is None:
            from ppasr_amd import _lib
            if not torch.cuda.is_available():
                raise _lib.PPASRHipError("no HIP device visible: the fbank front-end has no CPU fallback")
            self._lib = _lib.load()
            h = ctypes.c_void_p()
            with torch.cuda.device(self._device):
                _lib.check(self._lib.ppasr_fbank_create(self._sr, self._n_mels, 25.0, 10.0, ctypes.byref(h)))
            self._h = h
        return self._h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.ppasr_fbank_destroy(h)
            self._h = None

    @property
    def feature_dim(self):
        return self._n_mels

    @property
    def use_db_normalization(self):
        return bool(self._use_db)

    @property
    def target_db(self):
        return self._target_db

    def featurize_device(self, samples, sample_rate=None):
        """float32 mono samples in [-1, 1] (numpy or tensor) -> fbank [T, n_mels] float32 DEVICE tensor."""
        from ppasr_amd import _lib
        sr = sample_rate or self._sr
        if sr != self._sr:  # audio_featurizer.py:46-47: up / down-sample to the model's rate first (host side)
            samples = resample(np.asarray(torch.as_tensor(samples).cpu(), np.float32).reshape(-1), sr, self._sr)
        h = self._handle()
        x = torch.as_tensor(samples, dtype=torch.float32).reshape(-1).to(self._device).contiguous()
        n = int(x.numel())
        frames = int(self._lib.ppasr_fbank_frames(h, n))
        feats = torch.empty(frames, self._n_mels, dtype=torch.float32, device=self._device)
        if frames == 0:
            return feats
        need = int(self._lib.ppasr_fbank_workspace_bytes(h, n))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self._device)
        with torch.cuda.device(self._device):
            stream = torch.cuda.current_stream(self._device).cuda_stream
            _lib.check(self._lib.ppasr_fbank_compute(h, x.data_ptr(), n, int(bool(self._use_db)), float(self._target_db),
                                                     feats.data_ptr(), self._ws.data_ptr(), self._ws.numel(), stream))
            torch.cuda.current_stream(self._device).synchronize()  # `x` must outlive the kernel
        if self._use_db:  # the workspace's tail: [.. chunk sums .., gain, gain in dB] (csrc/fbank.hip k_gain)
            chunks = (n + 8191) // 8192
            self.last_gain, gain_db = (float(v) for v in self._ws[4 * chunks:4 * chunks + 8].view(torch.float32).cpu())
            check_gain_db(gain_db, self._target_db)
        return feats

    def featurize(self, samples, sample_rate=None):
        """float32 mono samples -> fbank [T, n_mels] float32 (numpy), the reference's return type."""
        return self.featurize_device(samples, sample_rate).cpu().numpy()


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
