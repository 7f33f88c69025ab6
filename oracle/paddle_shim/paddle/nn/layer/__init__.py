from . import conv  # noqa: F401
