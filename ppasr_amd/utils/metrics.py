"""Host-side mirror of ``ppasr/utils/metrics.py`` (``cer`` :4-13, ``wer`` :16-29) and of ``labels_to_string``
(``ppasr/utils/utils.py:59-65``).  The reference calls the ``Levenshtein`` C extension (not installable offline);
the edit distance here is the classic two-row dynamic programme (insert / delete / substitute, unit costs), which is
what ``Levenshtein.distance`` computes -- as host code of the C library, like the reference's C extension."""

__all__ = ["edit_distance", "cer", "wer", "labels_to_string"]


def edit_distance(a, b):
    """Levenshtein distance between two sequences (strings or lists of hashable items): the two-row dynamic programme
    in the C library (``ppasr_edit_distance``, host code)."""
    import numpy as np

    from ppasr_amd import _lib
    if isinstance(a, str) and isinstance(b, str):
        ia = np.frombuffer(a.encode("utf-32-le"), dtype=np.int32)
        ib = np.frombuffer(b.encode("utf-32-le"), dtype=np.int32)
    else:
        ids = {}
        ia = np.array([ids.setdefault(x, len(ids)) for x in a], dtype=np.int32)
        ib = np.array([ids.setdefault(x, len(ids)) for x in b], dtype=np.int32)
    d = _lib.load().ppasr_edit_distance(ia.ctypes.data if len(ia) else None, len(ia), ib.ctypes.data if len(ib) else None,
                                        len(ib))
    if d < 0:
        raise ValueError("edit_distance: bad arguments")
    return int(d)


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
