"""ORACLE (test infrastructure) -- `soundfile` stand-in: the reference imports it at module load (data_utils/audio.py,
data_utils/utils.py).  `read` covers what tests/golden/make_wav_goldens.py needs: PCM16 .wav files / file objects through
the standard library's `wave` (libsndfile's float32 conversion of PCM16 is x / 32768)."""
import wave as _wave

import numpy as _np


def read(file, dtype="float32", **kw):
    with _wave.open(file, "rb") as w:
        sr, ch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw != 2:
        raise NotImplementedError("soundfile shim: PCM16 only")
    x = _np.frombuffer(raw, dtype=_np.int16)
    if ch > 1:
        x = x.reshape(-1, ch)
    if dtype == "float32":
        return x.astype(_np.float32) / _np.float32(32768.0), sr
    if dtype == "int16":
        return x.copy(), sr
    raise NotImplementedError(dtype)
