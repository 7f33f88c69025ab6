"""ORACLE (test infrastructure) -- ``paddle.nn.initializer`` subset.  Values only matter for shapes: every
fixture loads a seeded state dict by name afterwards."""
import math

import torch as _torch


class Initializer:
    def __call__(self, t):
        raise NotImplementedError


def _fans(t):
    shp = list(t.shape)
    if len(shp) == 0:
        return 1, 1
    if len(shp) == 1:
        return shp[0], shp[0]
    if len(shp) == 2:  # paddle Linear weights are [in, out]
        return shp[0], shp[1]
    rf = 1
    for s in shp[2:]:
        rf *= s
    return shp[1] * rf, shp[0] * rf


class Constant(Initializer):
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, t):
        t.fill_(self.value)


class Uniform(Initializer):
    def __init__(self, low=-1.0, high=1.0, name=None):
        self.low, self.high = low, high

    def __call__(self, t):
        t.uniform_(self.low, self.high)


class Normal(Initializer):
    def __init__(self, mean=0.0, std=1.0, name=None):
        self.mean, self.std = mean, std

    def __call__(self, t):
        t.normal_(self.mean, self.std)


class XavierUniform(Initializer):
    def __init__(self, fan_in=None, fan_out=None, name=None):
        self.fan_in, self.fan_out = fan_in, fan_out

    def __call__(self, t):
        fi, fo = _fans(t)
        fi = self.fan_in or fi
        fo = self.fan_out or fo
        b = math.sqrt(6.0 / (fi + fo))
        t.uniform_(-b, b)


class KaimingUniform(Initializer):
    def __init__(self, fan_in=None, negative_slope=0.0, nonlinearity="relu"):
        self.fan_in, self.negative_slope, self.nonlinearity = fan_in, negative_slope, nonlinearity

    def __call__(self, t):
        fi = self.fan_in or _fans(t)[0]
        gain = math.sqrt(2.0 / (1 + self.negative_slope ** 2)) if self.nonlinearity == "leaky_relu" else math.sqrt(2.0)
        b = gain * math.sqrt(3.0 / fi)
        t.uniform_(-b, b)


class KaimingNormal(KaimingUniform):
    def __call__(self, t):
        fi = self.fan_in or _fans(t)[0]
        t.normal_(0.0, math.sqrt(2.0 / fi))
