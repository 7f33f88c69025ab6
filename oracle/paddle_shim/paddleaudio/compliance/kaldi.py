"""ORACLE (test infrastructure) -- ``paddleaudio.compliance.kaldi.fbank`` with the signature the reference calls
(ppasr/data_utils/featurizer/audio_featurizer.py:120-138).  paddleaudio (requirements.txt:14, ``paddleaudio>=1.0.1``) is
NOT in /root/reference and not installable offline: the arithmetic here is oracle/fbank_oracle.py's float64 restatement of
Kaldi's published algorithm with that function's defaults -- PARITY UNPINNED, like the oracle it forwards to.  It exists so
that the reference's own ``AudioFeaturizer`` / ``PPASRPredictor`` source can run end to end in
tests/golden/make_wav_goldens.py."""
import numpy as _np

import paddle as _paddle


def fbank(waveform, n_mels=23, frame_length=25, frame_shift=10, dither=0.0, sr=16000, **kw):
    if dither != 0.0:
        raise NotImplementedError("paddle shim: fbank with dither (training only)")
    if kw:
        raise NotImplementedError(f"paddle shim: fbank options {sorted(kw)}")
    from oracle.fbank_oracle import kaldi_fbank
    x = waveform.numpy()
    assert x.ndim == 2 and x.shape[0] == 1, x.shape  # (channel, time); the reference passes the int16-scaled samples
    return _paddle.to_tensor(kaldi_fbank(x[0], sr=sr, n_mels=n_mels, frame_length_ms=float(frame_length),
                                         frame_shift_ms=float(frame_shift)).astype(_np.float32))


def mfcc(*a, **k):
    raise NotImplementedError("stub: paddleaudio.compliance.kaldi.mfcc")
