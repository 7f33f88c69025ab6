"""ORACLE (test infrastructure) -- import-only stub: the fbank arithmetic of paddleaudio is NOT provided here
(oracle/fbank_oracle.py is a separate restatement)."""


def fbank(*a, **k):
    raise NotImplementedError("stub: paddleaudio.compliance.kaldi.fbank")


def mfcc(*a, **k):
    raise NotImplementedError("stub: paddleaudio.compliance.kaldi.mfcc")
