"""ORACLE (test infrastructure) -- ``typeguard.typechecked`` as the identity decorator (the reference decorates
constructors with it; the checks themselves are not part of the computation)."""


def typechecked(func=None, **kw):
    if func is None:
        return lambda f: f
    return func


check_argument_types = lambda *a, **k: True  # noqa: E731
check_return_type = lambda *a, **k: True  # noqa: E731
