"""CPU tests of the host-side metrics (ppasr/utils/metrics.py restated without the Levenshtein extension)."""
import itertools

import numpy as np
import pytest

from ppasr_amd.utils.metrics import cer, edit_distance, labels_to_string, wer


def _brute(a, b):
    # exhaustive recursion (tiny inputs only)
    if not a:
        return len(b)
    if not b:
        return len(a)
    return min(_brute(a[1:], b) + 1, _brute(a, b[1:]) + 1, _brute(a[1:], b[1:]) + (a[0] != b[0]))


def test_edit_distance_known_answers():
    assert edit_distance("kitten", "sitting") == 3
    assert edit_distance("flaw", "lawn") == 2
    assert edit_distance("", "abc") == 3 and edit_distance("abc", "") == 3 and edit_distance("", "") == 0
    assert edit_distance("今天天气很好", "今天天很好啊") == 2
    rng = np.random.Generator(np.random.PCG64(0))
    for _ in range(60):
        a = "".join(rng.choice(list("abc"), size=rng.integers(0, 6)))
        b = "".join(rng.choice(list("abc"), size=rng.integers(0, 6)))
        assert edit_distance(a, b) == _brute(a, b) == edit_distance(b, a)


def test_cer_wer_follow_the_reference_conventions():
    assert cer("a b c", "abc") == 0.0                       # spaces dropped (metrics.py:12)
    assert cer("abd", "abc") == pytest.approx(1 / 3)
    assert wer("the cat sat", "the cat sat") == 0.0
    assert wer("the cat", "the cat sat") == pytest.approx(1 / 3)
    assert wer("a cat sat down", "the cat sat") == pytest.approx(2 / 3)
    with pytest.raises(ZeroDivisionError):
        cer("abc", "")


def test_labels_to_string():
    vocab = ["<blank>", "<unk>", "a", "b", "<space>", "<eos>"]
    labels = np.array([[2, 4, 3, 5, -1], [1, 2, 0, 3, -1]])
    assert labels_to_string(labels, vocab, eos=5) == ["a b", "ab"]


def test_resample_keeps_a_tone_and_the_duration():
    """Host-side resampler behind AudioFeaturizer (the reference resamples to the model's rate first,
    audio_featurizer.py:46-47): 8 kHz -> 16 kHz and 44.1 kHz -> 16 kHz keep a 440 Hz tone's frequency and level."""
    import numpy as np

    from ppasr_amd.data_utils.featurizer import resample
    for sr in (8000, 44100, 16000):
        t = np.arange(sr) / sr
        x = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
        y = resample(x, sr, 16000)
        assert y.dtype == np.float32 and abs(len(y) - 16000) <= 1
        spec = np.abs(np.fft.rfft(y[2000:14000] * np.hanning(12000)))
        assert abs(np.argmax(spec) * 16000 / 12000 - 440) < 2.0
        assert abs(np.sqrt(np.mean(y[2000:14000] ** 2)) - 0.5 / np.sqrt(2)) < 5e-3
