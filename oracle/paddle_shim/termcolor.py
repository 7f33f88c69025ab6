"""ORACLE (test infrastructure) -- import-only stub of ``termcolor`` (ppasr/utils/logger.py)."""


def colored(text, *a, **k):
    return text
