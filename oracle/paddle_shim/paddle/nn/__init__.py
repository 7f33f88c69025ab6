"""ORACLE (test infrastructure) -- ``paddle.nn`` subset on torch CPU tensors.  See ../../README.md."""
import collections
import math

import torch as _torch

import paddle
from paddle import Parameter, Tensor

from . import functional  # noqa: F401
from . import functional as F
from . import initializer  # noqa: F401
from . import initializer as I


# --------------------------------------------------------------------------- Layer
class Layer:
    """``paddle.nn.Layer``: parameter / buffer / sub-layer registries with Paddle's dotted
    ``state_dict`` names (parameters, then persistable buffers, then sub-layers)."""

    def __init__(self, name_scope=None, dtype="float32"):
        object.__setattr__(self, "_parameters", collections.OrderedDict())
        object.__setattr__(self, "_buffers", collections.OrderedDict())
        object.__setattr__(self, "_non_persistable", set())
        object.__setattr__(self, "_sub_layers", collections.OrderedDict())
        object.__setattr__(self, "training", True)
        object.__setattr__(self, "_dtype", dtype)

    # ---- attribute plumbing
    def __setattr__(self, name, value):
        d = self.__dict__
        if "_parameters" not in d:
            raise RuntimeError("paddle shim: Layer.__init__() must run before assigning attributes")
        if isinstance(value, Parameter):
            d.pop(name, None)
            self._sub_layers.pop(name, None)
            self._parameters[name] = value
        elif isinstance(value, Layer):
            d.pop(name, None)
            self._parameters.pop(name, None)
            self._sub_layers[name] = value
        elif name in self._parameters:
            if value is not None:
                raise TypeError(f"paddle shim: cannot assign {type(value)} to parameter {name}")
            self._parameters[name] = None
        elif name in self._sub_layers:
            del self._sub_layers[name]
            object.__setattr__(self, name, value)
        elif name in self._buffers:
            self._buffers[name] = value
        else:
            object.__setattr__(self, name, value)

    def __getattr__(self, name):
        d = self.__dict__
        for reg in ("_parameters", "_sub_layers", "_buffers"):
            if reg in d and name in d[reg]:
                return d[reg][name]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __delattr__(self, name):
        for reg in (self._parameters, self._sub_layers, self._buffers):
            if name in reg:
                del reg[name]
                return
        object.__delattr__(self, name)

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    # ---- registration
    def create_parameter(self, shape, attr=None, dtype=None, is_bias=False, default_initializer=None):
        if attr is False:
            return None
        init = None
        trainable = True
        if isinstance(attr, paddle.ParamAttr):
            init = attr.initializer
            trainable = attr.trainable
        elif isinstance(attr, I.Initializer):
            init = attr
        if init is None:
            init = default_initializer
        if init is None:
            init = I.Constant(0.0) if is_bias else I.XavierUniform()
        t = _torch.empty([int(s) for s in shape], dtype=paddle._dtype(dtype) or paddle.float32)
        init(t)
        return Parameter(t, trainable=trainable)

    def add_parameter(self, name, parameter):
        if parameter is not None and not isinstance(parameter, Parameter):
            raise TypeError("paddle shim: add_parameter needs a Parameter")
        self.__dict__.pop(name, None)
        self._parameters[name] = parameter
        return parameter

    def add_sublayer(self, name, sublayer):
        self.__dict__.pop(name, None)
        self._sub_layers[str(name)] = sublayer
        return sublayer

    def register_buffer(self, name, tensor, persistable=True):
        self.__dict__.pop(name, None)
        self._buffers[name] = tensor
        if not persistable:
            self._non_persistable.add(name)

    # ---- traversal
    def named_sublayers(self, prefix="", include_self=False):
        if include_self:
            yield prefix, self
        for n, l in self._sub_layers.items():
            if l is None:
                continue
            p = prefix + ("." if prefix else "") + n
            yield p, l
            yield from l.named_sublayers(prefix=p)

    def sublayers(self, include_self=False):
        return [l for _, l in self.named_sublayers(include_self=include_self)]

    def children(self):
        return [l for l in self._sub_layers.values() if l is not None]

    def named_parameters(self, prefix="", include_sublayers=True):
        seen = set()
        for lp, layer in self.named_sublayers(prefix=prefix, include_self=True):
            for n, p in layer._parameters.items():
                if p is None or id(p) in seen:
                    continue
                seen.add(id(p))
                yield lp + ("." if lp else "") + n, p
            if not include_sublayers:
                break

    def parameters(self, include_sublayers=True):
        return [p for _, p in self.named_parameters(include_sublayers=include_sublayers)]

    def state_dict(self, destination=None, include_sublayers=True, structured_name_prefix="", use_hook=True):
        # paddle Layer._state_dict_impl: own parameters, own persistable buffers, then sub-layers; no de-duplication
        out = collections.OrderedDict() if destination is None else destination
        for n, p in self._parameters.items():
            if p is not None:
                out[structured_name_prefix + n] = p
        for n, b in self._buffers.items():
            if b is not None and n not in self._non_persistable:
                out[structured_name_prefix + n] = b
        if include_sublayers:
            for n, l in self._sub_layers.items():
                if l is not None:
                    l.state_dict(out, True, structured_name_prefix + n + ".")
        return out

    def set_state_dict(self, state_dict, use_structured_name=True):
        own = self.state_dict()
        missing, unexpected = [], []
        for k, dst in own.items():
            if k not in state_dict:
                missing.append(k)
                continue
            dst.set_value(state_dict[k])
        for k in state_dict:
            if k not in own:
                unexpected.append(k)
        return missing, unexpected

    set_dict = set_state_dict
    load_dict = set_state_dict

    def eval(self):
        for l in self.sublayers(include_self=True):
            object.__setattr__(l, "training", False)
        return self

    def train(self):
        for l in self.sublayers(include_self=True):
            object.__setattr__(l, "training", True)
        return self

    def full_name(self):
        return type(self).__name__.lower()

    def to(self, *a, **k):
        return self

    def apply(self, fn):
        for l in self.sublayers(include_self=True):
            fn(l)
        return self


class LayerList(Layer):
    def __init__(self, sublayers=None):
        super().__init__()
        if sublayers is not None:
            for i, l in enumerate(sublayers):
                self.add_sublayer(str(i), l)

    def _abs(self, idx):
        n = len(self)
        if not -n <= idx < n:
            raise IndexError(idx)
        return idx % n

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return LayerList(list(self._sub_layers.values())[idx])
        return self._sub_layers[str(self._abs(int(idx)))]

    def __setitem__(self, idx, layer):
        self._sub_layers[str(self._abs(int(idx)))] = layer

    def __len__(self):
        return len(self._sub_layers)

    def __iter__(self):
        return iter(self._sub_layers.values())

    def append(self, sublayer):
        self.add_sublayer(str(len(self)), sublayer)
        return self

    def extend(self, sublayers):
        for l in sublayers:
            self.append(l)
        return self


class Sequential(Layer):
    def __init__(self, *layers):
        super().__init__()
        if len(layers) == 1 and isinstance(layers[0], (list, tuple)) and layers[0] and isinstance(layers[0][0], (list, tuple)):
            layers = layers[0]
        for i, l in enumerate(layers):
            if isinstance(l, (list, tuple)):
                self.add_sublayer(l[0], l[1])
            else:
                self.add_sublayer(str(i), l)

    def __getitem__(self, idx):
        if isinstance(idx, str):
            return self._sub_layers[idx]
        return list(self._sub_layers.values())[idx]

    def __len__(self):
        return len(self._sub_layers)

    def __iter__(self):
        return iter(self._sub_layers.values())

    def forward(self, x):
        for l in self._sub_layers.values():
            x = l(x)
        return x


# --------------------------------------------------------------------------- dense / conv / norm
class Linear(Layer):
    """``y = x @ weight[in, out] + bias``"""

    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        self.weight = self.create_parameter([in_features, out_features], attr=weight_attr)
        b = self.create_parameter([out_features], attr=bias_attr, is_bias=True)
        if b is None:
            self.bias = None
        else:
            self.bias = b
        self.name = name

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class Identity(Layer):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x):
        return x


class Dropout(Layer):
    def __init__(self, p=0.5, axis=None, mode="upscale_in_train", name=None):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.training and self.p > 0:
            raise NotImplementedError("paddle shim: Dropout in training mode (call model.eval())")
        return x


class Embedding(Layer):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, sparse=False, weight_attr=None, name=None):
        super().__init__()
        self.weight = self.create_parameter([num_embeddings, embedding_dim], attr=weight_attr)
        self._padding_idx = padding_idx

    def forward(self, x):
        return Tensor(self.weight._t[x._t.long()])


class LayerNorm(Layer):
    def __init__(self, normalized_shape, epsilon=1e-05, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = [normalized_shape]
        self._normalized_shape = list(normalized_shape)
        self._epsilon = epsilon
        w = self.create_parameter(self._normalized_shape, attr=weight_attr, default_initializer=I.Constant(1.0))
        b = self.create_parameter(self._normalized_shape, attr=bias_attr, is_bias=True)
        self.weight = w
        self.bias = b

    def forward(self, x):
        return F.layer_norm(x, self._normalized_shape, self.weight, self.bias, self._epsilon)


class _BatchNormBase(Layer):
    def __init__(self, num_features, momentum=0.9, epsilon=1e-05, weight_attr=None, bias_attr=None, data_format="NCL",
                 use_global_stats=None, name=None):
        super().__init__()
        self._epsilon = epsilon
        self._data_format = data_format
        self.weight = self.create_parameter([num_features], attr=weight_attr, default_initializer=I.Constant(1.0))
        self.bias = self.create_parameter([num_features], attr=bias_attr, is_bias=True)
        self.register_buffer("_mean", paddle.zeros([num_features]))
        self.register_buffer("_variance", paddle.ones([num_features]))

    def forward(self, x):
        if self.training:
            raise NotImplementedError("paddle shim: BatchNorm in training mode")
        t = x._t
        ch = 1 if self._data_format in ("NCL", "NCHW", "NC") else t.dim() - 1
        shp = [1] * t.dim()
        shp[ch] = -1
        y = (t - self._mean._t.reshape(shp)) / _torch.sqrt(self._variance._t.reshape(shp) + self._epsilon)
        return Tensor(y * self.weight._t.reshape(shp) + self.bias._t.reshape(shp))


class BatchNorm1D(_BatchNormBase):
    pass


class BatchNorm2D(_BatchNormBase):
    def __init__(self, num_features, momentum=0.9, epsilon=1e-05, weight_attr=None, bias_attr=None, data_format="NCHW",
                 use_global_stats=None, name=None):
        super().__init__(num_features, momentum, epsilon, weight_attr, bias_attr, data_format, use_global_stats, name)


from .layer.conv import _ConvNd  # noqa: E402


class Conv1D(_ConvNd):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 padding_mode="zeros", weight_attr=None, bias_attr=None, data_format="NCL"):
        super().__init__(in_channels, out_channels, kernel_size, False, 1, stride=stride, padding=padding,
                         padding_mode=padding_mode, dilation=dilation, groups=groups, weight_attr=weight_attr,
                         bias_attr=bias_attr, data_format=data_format)

    def forward(self, x):
        return F.conv1d(x, self.weight, self.bias, self._stride, self._padding, self._dilation, self._groups,
                        self._data_format)


class Conv2D(_ConvNd):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 padding_mode="zeros", weight_attr=None, bias_attr=None, data_format="NCHW"):
        super().__init__(in_channels, out_channels, kernel_size, False, 2, stride=stride, padding=padding,
                         padding_mode=padding_mode, dilation=dilation, groups=groups, weight_attr=weight_attr,
                         bias_attr=bias_attr, data_format=data_format)

    def forward(self, x):
        return F.conv2d(x, self.weight, self.bias, self._stride, self._padding, self._dilation, self._groups,
                        self._data_format)


class AvgPool1D(Layer):
    def __init__(self, kernel_size, stride=None, padding=0, exclusive=True, ceil_mode=False, name=None):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.exclusive, self.ceil_mode = exclusive, ceil_mode

    def forward(self, x):
        return F.avg_pool1d(x, self.kernel_size, self.stride, self.padding, self.exclusive, self.ceil_mode)


# --------------------------------------------------------------------------- activations
def _act(name, fn):
    def __init__(self, *args, **kwargs):
        Layer.__init__(self)
        self._args, self._kwargs = args, kwargs

    def forward(self, x):
        return Tensor(fn(x._t, *self._args, **self._kwargs))

    return type(name, (Layer,), {"__init__": __init__, "forward": forward})


ReLU = _act("ReLU", lambda t: _torch.relu(t))
ReLU6 = _act("ReLU6", lambda t: _torch.clamp(t, 0.0, 6.0))
Sigmoid = _act("Sigmoid", lambda t: _torch.sigmoid(t))
Tanh = _act("Tanh", lambda t: _torch.tanh(t))
Swish = _act("Swish", lambda t: t * _torch.sigmoid(t))
Silu = Swish
GELU = _act("GELU", lambda t, approximate=False: _torch.nn.functional.gelu(t, approximate="tanh" if approximate else "none"))
ELU = _act("ELU", lambda t, alpha=1.0: _torch.nn.functional.elu(t, alpha))
SELU = _act("SELU", lambda t: _torch.selu(t))
LeakyReLU = _act("LeakyReLU", lambda t, negative_slope=0.01: _torch.nn.functional.leaky_relu(t, negative_slope))
Hardtanh = _act("Hardtanh", lambda t, min=-1.0, max=1.0: _torch.clamp(t, min, max))  # noqa: A002
Hardswish = _act("Hardswish", lambda t: _torch.nn.functional.hardswish(t))
Hardshrink = _act("Hardshrink", lambda t, threshold=0.5: _torch.nn.functional.hardshrink(t, threshold))


class Softmax(Layer):
    def __init__(self, axis=-1, name=None):
        super().__init__()
        self._axis = axis

    def forward(self, x):
        return F.softmax(x, self._axis)


# --------------------------------------------------------------------------- losses (constructible; training is out of scope)
class CTCLoss(Layer):
    def __init__(self, blank=0, reduction="mean"):
        super().__init__()

    def forward(self, *a, **k):
        raise NotImplementedError("paddle shim: training losses are out of scope")


class KLDivLoss(CTCLoss):
    def __init__(self, reduction="mean"):
        Layer.__init__(self)


# --------------------------------------------------------------------------- recurrent layers
class _Cell(Layer):
    """Holder of one direction's parameters under Paddle's cell names (weight_ih, weight_hh, bias_ih, bias_hh)."""

    def __init__(self, input_size, hidden_size, gates):
        super().__init__()
        std = 1.0 / math.sqrt(hidden_size)
        u = I.Uniform(-std, std)
        self.weight_ih = self.create_parameter([gates * hidden_size, input_size], default_initializer=u)
        self.weight_hh = self.create_parameter([gates * hidden_size, hidden_size], default_initializer=u)
        self.bias_ih = self.create_parameter([gates * hidden_size], is_bias=True, default_initializer=u)
        self.bias_hh = self.create_parameter([gates * hidden_size], is_bias=True, default_initializer=u)


class _RNN(Layer):
    def __init__(self, cell):
        super().__init__()
        self.cell = cell


class _BiRNN(Layer):
    def __init__(self, cell_fw, cell_bw):
        super().__init__()
        self.cell_fw = cell_fw
        self.cell_bw = cell_bw


class _RNNBase(LayerList):
    """``paddle.nn.layer.rnn.RNNBase`` for ``num_layers == 1`` stacks as the reference builds them
    (deepspeech2/encoder.py:36-48).  Registers every parameter twice like Paddle does: as
    ``weight_ih_l0[_reverse]`` … on this layer and as ``0.cell[_fw|_bw].weight_ih`` … on the wrapped cells."""

    GATES = None

    def __init__(self, input_size, hidden_size, num_layers=1, direction="forward", time_major=False, dropout=0.0,
                 weight_ih_attr=None, weight_hh_attr=None, bias_ih_attr=None, bias_hh_attr=None, name=None):
        super().__init__()
        if num_layers != 1:
            raise NotImplementedError("paddle shim: RNN num_layers != 1")
        if time_major:
            raise NotImplementedError("paddle shim: RNN time_major")
        if direction in ("forward",):
            self.num_directions = 1
            cells = [_Cell(input_size, hidden_size, self.GATES)]
            self.append(_RNN(cells[0]))
        elif direction in ("bidirect", "bidirectional"):
            self.num_directions = 2
            cells = [_Cell(input_size, hidden_size, self.GATES), _Cell(input_size, hidden_size, self.GATES)]
            self.append(_BiRNN(cells[0], cells[1]))
        else:
            raise ValueError(direction)
        object.__setattr__(self, "_cells", cells)
        self.hidden_size = hidden_size
        self.input_size = input_size
        self.num_layers = num_layers
        for d, cell in enumerate(cells):
            sfx = "_reverse" if d == 1 else ""
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                setattr(self, f"{n}_l0{sfx}", getattr(cell, n))

    def _step(self, cell, x_t, state):
        raise NotImplementedError

    def _init_state(self, B, dtype):
        raise NotImplementedError

    def _run_direction(self, cell, x, state, lens, reverse):
        """x [B, T, I] torch; state tuple of [B, H]; lens [B] long or None.  Outputs at t >= len are 0, the state
        stops at len; the reverse direction walks len-1 .. 0 of every sequence."""
        B, T, _ = x.shape
        out = _torch.zeros(B, T, self.hidden_size, dtype=x.dtype)
        steps = range(T - 1, -1, -1) if reverse else range(T)
        for t in steps:
            new_state, h = self._step(cell, x[:, t], state)
            if lens is None:
                state = new_state
                out[:, t] = h
            else:
                m = (t < lens).to(x.dtype).unsqueeze(1)  # [B, 1]
                state = tuple(m * n + (1 - m) * s for n, s in zip(new_state, state))
                out[:, t] = m * h
        return out, state

    def forward(self, inputs, initial_states=None, sequence_length=None):
        x = inputs._t
        B = x.shape[0]
        lens = None if sequence_length is None else sequence_length._t.to(_torch.int64)
        init = self._unpack_states(initial_states, B, x.dtype)
        outs, finals = [], []
        for d, cell in enumerate(self._cells):
            o, s = self._run_direction(cell, x, init[d], lens, reverse=(d == 1))
            outs.append(o)
            finals.append(s)
        y = outs[0] if len(outs) == 1 else _torch.cat(outs, dim=-1)
        return Tensor(y), self._pack_states(finals)


class LSTM(_RNNBase):
    GATES = 4

    def _unpack_states(self, st, B, dtype):
        if st is None:
            z = _torch.zeros(B, self.hidden_size, dtype=dtype)
            return [(z, z)] * self.num_directions
        h, c = st
        return [(h._t[d], c._t[d]) for d in range(self.num_directions)]

    def _pack_states(self, finals):
        h = _torch.stack([f[0] for f in finals], 0)
        c = _torch.stack([f[1] for f in finals], 0)
        return Tensor(h), Tensor(c)

    def _step(self, cell, x_t, state):
        # paddle LSTMCell.forward: gates = x W_ih^T + b_ih + h W_hh^T + b_hh; chunks i, f, g, o
        h, c = state
        g = x_t @ cell.weight_ih._t.t() + cell.bias_ih._t + h @ cell.weight_hh._t.t() + cell.bias_hh._t
        i, f, cand, o = g.chunk(4, dim=-1)
        c2 = _torch.sigmoid(f) * c + _torch.sigmoid(i) * _torch.tanh(cand)
        h2 = _torch.sigmoid(o) * _torch.tanh(c2)
        return (h2, c2), h2


class GRU(_RNNBase):
    GATES = 3

    def _unpack_states(self, st, B, dtype):
        if st is None:
            z = _torch.zeros(B, self.hidden_size, dtype=dtype)
            return [(z,)] * self.num_directions
        return [(st._t[d],) for d in range(self.num_directions)]

    def _pack_states(self, finals):
        return Tensor(_torch.stack([f[0] for f in finals], 0))

    def _step(self, cell, x_t, state):
        # paddle GRUCell.forward: r, z, c chunks; c = tanh(x_c + r * h_c); h = (h_prev - c) * z + c
        (h,) = state
        xg = x_t @ cell.weight_ih._t.t() + cell.bias_ih._t
        hg = h @ cell.weight_hh._t.t() + cell.bias_hh._t
        x_r, x_z, x_c = xg.chunk(3, dim=-1)
        h_r, h_z, h_c = hg.chunk(3, dim=-1)
        r = _torch.sigmoid(x_r + h_r)
        z = _torch.sigmoid(x_z + h_z)
        c = _torch.tanh(x_c + r * h_c)
        h2 = (h - c) * z + c
        return (h2,), h2
