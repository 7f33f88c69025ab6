"""ORACLE (test infrastructure) -- ``paddle.nn.functional`` subset on torch CPU tensors.  See ../../README.md."""
import numpy as _np
import torch as _torch
import torch.nn.functional as _TF

import paddle
from paddle import Tensor, _int, _raw


def linear(x, weight, bias=None, name=None):
    y = _torch.matmul(x._t, weight._t)  # weight is [in, out]
    if bias is not None:
        y = y + bias._t
    return Tensor(y)


def softmax(x, axis=-1, dtype=None, name=None):
    t = x._t if dtype is None else x._t.to(paddle._dtype(dtype))
    return Tensor(_torch.softmax(t, dim=_int(axis)))


def log_softmax(x, axis=-1, dtype=None, name=None):
    t = x._t if dtype is None else x._t.to(paddle._dtype(dtype))
    return Tensor(_torch.log_softmax(t, dim=_int(axis)))


def relu(x, name=None):
    return Tensor(_torch.relu(x._t))


def sigmoid(x, name=None):
    return Tensor(_torch.sigmoid(x._t))


def swish(x, name=None):
    return Tensor(x._t * _torch.sigmoid(x._t))


silu = swish


def glu(x, axis=-1, name=None):
    a, b = _torch.chunk(x._t, 2, dim=_int(axis))
    return Tensor(a * _torch.sigmoid(b))


def dropout(x, p=0.5, axis=None, training=True, mode="upscale_in_train", name=None):
    if training and p > 0:
        raise NotImplementedError("paddle shim: dropout with p > 0 in training mode")
    return x


def one_hot(x, num_classes, name=None):
    return Tensor(_TF.one_hot(x._t.long(), num_classes).to(_torch.float32))


def layer_norm(x, normalized_shape, weight=None, bias=None, epsilon=1e-05, name=None):
    if isinstance(normalized_shape, int):
        normalized_shape = [normalized_shape]
    n = len(normalized_shape)
    t = x._t
    dims = list(range(t.dim() - n, t.dim()))
    mean = t.mean(dim=dims, keepdim=True)
    var = ((t - mean) ** 2).mean(dim=dims, keepdim=True)  # biased
    y = (t - mean) / _torch.sqrt(var + epsilon)
    if weight is not None:
        y = y * weight._t
    if bias is not None:
        y = y + bias._t
    return Tensor(y)


def _tuple(v, n):
    if isinstance(v, (int, _np.integer)):
        return (int(v),) * n
    v = [_int(a) for a in v]
    if len(v) == 1:
        v = v * n
    return tuple(v)


def _conv_padding(padding, n):
    """int | [p]*n -> symmetric per-dim; [l0, r0, l1, r1 ...] -> explicit pairs (applied with a zero pad)."""
    if isinstance(padding, str):
        raise NotImplementedError("paddle shim: string conv padding")
    if isinstance(padding, (int, _np.integer)):
        return (int(padding),) * n, None
    p = [_int(a) for a in padding]
    if len(p) == n:
        return tuple(p), None
    if len(p) == 2 * n:
        return (0,) * n, p
    raise ValueError(f"paddle shim: conv padding {padding}")


def _conv(x, weight, bias, stride, padding, dilation, groups, n, channel_last):
    t = x._t
    if channel_last:
        t = t.movedim(-1, 1)
    sym, explicit = _conv_padding(padding, n)
    if explicit is not None:
        # pairs are ordered first spatial dim first; torch pad wants the last dim first
        pads = []
        for i in reversed(range(n)):
            pads += [explicit[2 * i], explicit[2 * i + 1]]
        t = _TF.pad(t, pads)
    fn = _TF.conv1d if n == 1 else _TF.conv2d
    y = fn(t, weight._t, None if bias is None else bias._t, _tuple(stride, n), sym, _tuple(dilation, n), int(groups))
    if channel_last:
        y = y.movedim(1, -1)
    return Tensor(y)


def conv1d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, data_format="NCL", name=None):
    return _conv(x, weight, bias, stride, padding, dilation, groups, 1, data_format == "NLC")


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, data_format="NCHW", name=None):
    return _conv(x, weight, bias, stride, padding, dilation, groups, 2, data_format == "NHWC")


def pad(x, pad, mode="constant", value=0.0, data_format="NCHW", name=None):  # noqa: A002
    """paddle.nn.functional.pad (python/paddle/nn/functional/common.py):
    * a list/tuple of 2*ndim ints with mode 'constant': pairs (before, after) per dimension, dim 0 first;
    * otherwise (a shorter list, or a Tensor of any length): [left, right] / [left, right, top, bottom] on the
      spatial dims named by data_format -- the LAST spatial dim comes first."""
    if mode != "constant":
        raise NotImplementedError("paddle shim: pad mode " + mode)
    t = x._t
    nd = t.dim()
    is_tensor = isinstance(pad, Tensor)
    p = [int(v) for v in (pad._t.tolist() if is_tensor else [_int(a) for a in pad])]
    if not is_tensor and len(p) == 2 * nd:
        tp = []
        for i in reversed(range(nd)):
            tp += [p[2 * i], p[2 * i + 1]]
        return Tensor(_TF.pad(t, tp, value=value))
    data_format = data_format.upper()
    supported = {3: ("NCL", "NLC"), 4: ("NCHW", "NHWC"), 5: ("NCDHW", "NDHWC")}
    if nd not in supported:
        raise ValueError(f"paddle shim: pad on a {nd}-D tensor with {len(p)} pads")
    if data_format not in supported[nd]:
        # paddle asserts this; the reference relies on the default 'NCHW' for 4-D inputs only
        raise ValueError(f"paddle shim: data_format {data_format} invalid for {nd}-D input")
    if len(p) != 2 * (nd - 2):
        raise ValueError(f"paddle shim: {len(p)} pads for a {nd}-D input")
    channel_last = data_format in ("NLC", "NHWC", "NDHWC")
    if channel_last:
        t = t.movedim(-1, 1)
    y = _TF.pad(t, p, value=value)  # torch order == paddle order here: last spatial dim first
    if channel_last:
        y = y.movedim(1, -1)
    return Tensor(y)


def avg_pool1d(x, kernel_size, stride=None, padding=0, exclusive=True, ceil_mode=False, name=None):
    """[N, C, L]; exclusive=True divides each window by its number of in-range elements."""
    k = _tuple(kernel_size, 1)[0]
    s = k if stride is None else _tuple(stride, 1)[0]
    p = _tuple(padding, 1)[0]
    if p != 0:
        raise NotImplementedError("paddle shim: avg_pool1d padding")
    t = x._t
    L = t.shape[-1]
    if ceil_mode:
        n_out = (L - k + s - 1) // s + 1
    else:
        n_out = (L - k) // s + 1
    need = (n_out - 1) * s + k
    tp = _TF.pad(t, [0, max(0, need - L)])
    ones = _TF.pad(_torch.ones(L, dtype=t.dtype), [0, max(0, need - L)])
    win = tp.unfold(-1, k, s)[..., :n_out, :].sum(-1)
    cnt = ones.unfold(-1, k, s)[:n_out].sum(-1)
    return Tensor(win / (cnt if exclusive else float(k)))
