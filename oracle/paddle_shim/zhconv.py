"""ORACLE (test infrastructure) -- import-only stub of ``zhconv`` (data_utils/utils.py:14)."""


def convert(*a, **k):
    raise NotImplementedError("stub: zhconv.convert")
