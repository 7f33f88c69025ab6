"""ORACLE (test infrastructure) -- ``paddle.jit`` stand-ins: ``to_static`` returns the function unchanged
(the exported graph of trainer.py:675-682 computes what the dygraph method computes)."""


def to_static(function=None, input_spec=None, build_strategy=None, **kw):
    if function is None:
        return lambda f: f
    return function


def save(*a, **k):
    raise NotImplementedError("paddle shim: jit.save (no ProgramDesc here)")


def not_to_static(f):
    return f
