"""ORACLE (test infrastructure) -- ``paddle.static.InputSpec`` record."""


class InputSpec:
    def __init__(self, shape=None, dtype="float32", name=None, stop_gradient=False):
        self.shape, self.dtype, self.name = shape, dtype, name
