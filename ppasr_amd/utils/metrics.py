"""Host-side mirror of ``ppasr/utils/metrics.py`` (``cer`` :4-13, ``wer`` :16-29) and of ``labels_to_string``
(``ppasr/utils/utils.py:59-65``).  The reference calls the ``Levenshtein`` C extension (not installable offline);
the edit distance here is the classic two-row dynamic programme (insert / delete / substitute, unit costs), which is
what ``Levenshtein.distance`` computes."""

__all__ = ["edit_distance", "cer", "wer", "labels_to_string"]


def edit_distance(a, b):
    """Levenshtein distance between two sequences (strings or lists)."""
    if len(a) < len(b):
        a, b = b, a
    if len(b) == 0:
        return len(a)
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, cb in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb))
        prev = cur
    return prev[-1]


def cer(prediction, label):
    """Character error rate: spaces are dropped from both strings first (metrics.py:12); an empty label raises
    ZeroDivisionError like the reference."""
    prediction, label = prediction.replace(" ", ""), label.replace(" ", "")
    return edit_distance(prediction, label) / float(len(label))


def wer(prediction, label):
    """Word error rate: the reference maps every distinct word to one character and calls ``cer`` on the two
    strings (metrics.py:16-29), i.e. word-level edit distance / number of label words."""
    p, l = prediction.split(" "), label.split(" ")
    return edit_distance(p, l) / float(len(l))


def labels_to_string(label, vocabulary, eos, blank_index=0):
    out = []
    for row in label:
        ids = [int(i) for i in row if i != blank_index and i != -1 and i != eos]
        out.append("".join(vocabulary[i] for i in ids).replace("<space>", " ").replace("<unk>", ""))
    return out
