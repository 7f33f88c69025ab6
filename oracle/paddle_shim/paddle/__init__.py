"""ORACLE (test infrastructure) -- a minimal ``paddle`` backed by torch CPU tensors.

See ../README.md.  ``Tensor`` is an explicit wrapper around a ``torch.Tensor``: every operator
and method the reference's model files use is defined here with Paddle semantics, anything
else raises ``AttributeError``.  Not product code; never imported by ``ppasr_amd``.
"""
import builtins
import math as _math

import numpy as _np
import torch as _torch

__version__ = "2.5.1+torchshim"

# --------------------------------------------------------------------------- dtypes
float16 = _torch.float16
float32 = _torch.float32
float64 = _torch.float64
int8 = _torch.int8
uint8 = _torch.uint8
int16 = _torch.int16
int32 = _torch.int32
int64 = _torch.int64
bool = _torch.bool  # noqa: A001  (paddle.bool)

_DTYPE_BY_NAME = {"float16": float16, "float32": float32, "float64": float64, "int8": int8, "uint8": uint8,
                  "int16": int16, "int32": int32, "int64": int64, "bool": bool}
_default_dtype = float32


def _dtype(d):
    if d is None:
        return None
    if isinstance(d, _torch.dtype):
        return d
    if isinstance(d, str):
        return _DTYPE_BY_NAME[d]
    if isinstance(d, _np.dtype) or (isinstance(d, type) and issubclass(d, _np.generic)):
        return _DTYPE_BY_NAME[_np.dtype(d).name]
    raise TypeError(f"paddle shim: unknown dtype {d!r}")


def get_default_dtype():
    return "float32"


# --------------------------------------------------------------------------- Tensor
def _raw(x):
    """Tensor -> torch.Tensor; python scalars / numpy pass through."""
    return x._t if isinstance(x, Tensor) else x


def _int(x):
    if isinstance(x, Tensor):
        return builtins.int(x._t.item())
    return builtins.int(x)


def _shape(shape):
    if isinstance(shape, Tensor):
        return [builtins.int(v) for v in shape._t.tolist()]
    if isinstance(shape, (builtins.int, _np.integer)):
        return [builtins.int(shape)]
    return [_int(s) for s in shape]


def _index(key):
    if isinstance(key, tuple):
        return tuple(_index(k) for k in key)
    if isinstance(key, slice):
        f = lambda v: None if v is None else _int(v)  # noqa: E731
        return slice(f(key.start), f(key.stop), f(key.step))
    if isinstance(key, Tensor):
        t = key._t
        if t.dtype == _torch.bool or t.dim() > 0:
            return t
        return builtins.int(t.item())
    return key


class Tensor:
    """Wrapper with the subset of ``paddle.Tensor`` the reference models use."""

    __array_priority__ = 1000

    def __init__(self, t):
        assert isinstance(t, _torch.Tensor), type(t)
        self._t = t
        self.stop_gradient = True
        self.name = None

    # ---- meta
    @property
    def shape(self):
        return list(self._t.shape)

    @property
    def dtype(self):
        return self._t.dtype

    @property
    def ndim(self):
        return self._t.dim()

    @property
    def size(self):
        return self._t.numel()

    @property
    def T(self):
        return Tensor(self._t.permute(*reversed(range(self._t.dim()))))

    def dim(self):
        return self._t.dim()

    def numel(self):
        return self._t.numel()

    def numpy(self):
        return self._t.detach().cpu().numpy()

    def tolist(self):
        return self._t.tolist()

    def item(self, *args):
        if args:
            return self._t.flatten()[args[0]].item()
        return self._t.item()

    def clone(self):
        return Tensor(self._t.clone())

    def detach(self):
        return Tensor(self._t.detach())

    def cpu(self):
        return self

    def __repr__(self):
        return f"Tensor(shape={self.shape}, dtype={self.dtype},\n       {self._t})"

    def __len__(self):
        return self._t.shape[0]

    def __int__(self):
        return builtins.int(self._t.item())

    def __index__(self):
        return builtins.int(self._t.item())

    def __float__(self):
        return builtins.float(self._t.item())

    def __bool__(self):
        return builtins.bool(self._t.item())

    __hash__ = object.__hash__

    def __iter__(self):
        for i in range(self._t.shape[0]):
            yield Tensor(self._t[i])

    # ---- indexing
    def __getitem__(self, key):
        return Tensor(self._t[_index(key)])

    def __setitem__(self, key, value):
        self._t[_index(key)] = _raw(value)

    # ---- arithmetic (torch type promotion == paddle's for the float32/int cases on this path)
    def _bin(self, other, fn, reverse=False):
        o = _raw(other)
        if isinstance(o, _np.ndarray):
            o = _torch.from_numpy(o)
        return Tensor(fn(o, self._t) if reverse else fn(self._t, o))

    def __add__(self, o):
        return self._bin(o, _torch.add)

    def __radd__(self, o):
        return self._bin(o, _torch.add, True)

    def __sub__(self, o):
        return self._bin(o, _torch.sub)

    def __rsub__(self, o):
        return self._bin(o, _torch.sub, True)

    def __mul__(self, o):
        return self._bin(o, _torch.mul)

    def __rmul__(self, o):
        return self._bin(o, _torch.mul, True)

    def __truediv__(self, o):
        return self._bin(o, _torch.true_divide)

    def __rtruediv__(self, o):
        return self._bin(o, _torch.true_divide, True)

    def __floordiv__(self, o):
        return self._bin(o, lambda a, b: _torch.div(a, b, rounding_mode="floor"))

    def __rfloordiv__(self, o):
        return self._bin(o, lambda a, b: _torch.div(a, b, rounding_mode="floor"), True)

    def __mod__(self, o):
        return self._bin(o, _torch.remainder)

    def __pow__(self, o):
        return self._bin(o, _torch.pow)

    def __matmul__(self, o):
        return self._bin(o, _torch.matmul)

    def __neg__(self):
        return Tensor(-self._t)

    def __abs__(self):
        return Tensor(self._t.abs())

    def __invert__(self):
        return Tensor(~self._t)

    def __and__(self, o):
        return self._bin(o, lambda a, b: a & b)

    def __or__(self, o):
        return self._bin(o, lambda a, b: a | b)

    def __xor__(self, o):
        return self._bin(o, lambda a, b: a ^ b)

    def __eq__(self, o):  # elementwise, like paddle
        return self._bin(o, _torch.eq)

    def __ne__(self, o):
        return self._bin(o, _torch.ne)

    def __lt__(self, o):
        return self._bin(o, _torch.lt)

    def __le__(self, o):
        return self._bin(o, _torch.le)

    def __gt__(self, o):
        return self._bin(o, _torch.gt)

    def __ge__(self, o):
        return self._bin(o, _torch.ge)

    def __iadd__(self, o):
        self._t = self._t + _raw(o)
        return self

    def __isub__(self, o):
        self._t = self._t - _raw(o)
        return self

    def __imul__(self, o):
        self._t = self._t * _raw(o)
        return self

    # ---- methods (each mirrors the module-level function below)
    def astype(self, dtype):
        return Tensor(self._t.to(_dtype(dtype)))

    cast = astype

    def transpose(self, perm, name=None):
        return transpose(self, perm)

    def reshape(self, shape, name=None):
        return reshape(self, shape)

    def flatten(self, start_axis=0, stop_axis=-1):
        return Tensor(self._t.flatten(start_axis, stop_axis))

    def unsqueeze(self, axis, name=None):
        return unsqueeze(self, axis)

    def squeeze(self, axis=None, name=None):
        return squeeze(self, axis)

    def equal(self, y, name=None):
        return equal(self, y)

    def logical_and(self, y, name=None):
        return logical_and(self, y)

    def logical_or(self, y, name=None):
        return logical_or(self, y)

    def logical_not(self, name=None):
        return logical_not(self)

    def sum(self, axis=None, dtype=None, keepdim=False, name=None):
        return sum(self, axis, dtype, keepdim)

    def mean(self, axis=None, keepdim=False, name=None):
        return mean(self, axis, keepdim)

    def max(self, axis=None, keepdim=False, name=None):
        return max(self, axis, keepdim)

    def min(self, axis=None, keepdim=False, name=None):
        return min(self, axis, keepdim)

    def argmax(self, axis=None, keepdim=False, dtype="int64", name=None):
        return argmax(self, axis, keepdim, dtype)

    def all(self, axis=None, keepdim=False, name=None):
        if axis is None:
            return Tensor(self._t.all())
        return Tensor(self._t.all(dim=axis, keepdim=keepdim))

    def any(self, axis=None, keepdim=False, name=None):
        if axis is None:
            return Tensor(self._t.any())
        return Tensor(self._t.any(dim=axis, keepdim=keepdim))

    def expand(self, shape, name=None):
        return expand(self, shape)

    def broadcast_to(self, shape, name=None):
        return broadcast_to(self, shape)

    def repeat_interleave(self, repeats, axis=None, name=None):
        return repeat_interleave(self, repeats, axis)

    def masked_select(self, mask, name=None):
        return masked_select(self, mask)

    def split(self, num_or_sections, axis=0, name=None):
        return split(self, num_or_sections, axis)

    def matmul(self, y, transpose_x=False, transpose_y=False, name=None):
        return matmul(self, y, transpose_x, transpose_y)

    def exp(self, name=None):
        return exp(self)

    def log(self, name=None):
        return log(self)

    def sqrt(self, name=None):
        return sqrt(self)

    def abs(self, name=None):
        return Tensor(self._t.abs())

    def tanh(self, name=None):
        return tanh(self)

    def flip(self, axis, name=None):
        return flip(self, axis)

    def tile(self, repeat_times, name=None):
        return Tensor(self._t.repeat(*_shape(repeat_times)))

    def set_value(self, value):
        v = _raw(value)
        if isinstance(v, _np.ndarray):
            v = _torch.from_numpy(_np.ascontiguousarray(v))
        if list(v.shape) != list(self._t.shape):
            raise ValueError(f"paddle shim: set_value shape {list(v.shape)} != {list(self._t.shape)}")
        self._t = v.to(self._t.dtype).clone()


class Parameter(Tensor):
    def __init__(self, t, trainable=True):
        super().__init__(t)
        self.stop_gradient = not trainable
        self.trainable = trainable


def is_tensor(x):
    return isinstance(x, Tensor)


# --------------------------------------------------------------------------- creation
def to_tensor(data, dtype=None, place=None, stop_gradient=True):
    dt = _dtype(dtype)
    if isinstance(data, Tensor):
        t = data._t.clone()
    elif isinstance(data, _torch.Tensor):
        t = data.clone()
    elif isinstance(data, _np.ndarray):
        t = _torch.from_numpy(_np.ascontiguousarray(data)).clone()
    elif isinstance(data, _np.generic):
        t = _torch.from_numpy(_np.asarray(data)).clone()
    else:
        # python scalars / nested lists: float -> default dtype (float32), int -> int64, bool -> bool
        a = _np.asarray(data)
        if a.dtype == _np.float64:
            a = a.astype(_np.float32)
        t = _torch.from_numpy(_np.ascontiguousarray(a)).clone()
    if dt is not None:
        t = t.to(dt)
    return Tensor(t)


def zeros(shape, dtype=None, name=None):
    return Tensor(_torch.zeros(_shape(shape), dtype=_dtype(dtype) or _default_dtype))


def ones(shape, dtype=None, name=None):
    return Tensor(_torch.ones(_shape(shape), dtype=_dtype(dtype) or _default_dtype))


def empty(shape, dtype=None, name=None):
    return Tensor(_torch.zeros(_shape(shape), dtype=_dtype(dtype) or _default_dtype))


def full(shape, fill_value, dtype=None, name=None):
    return Tensor(_torch.full(_shape(shape), _raw(fill_value) if not isinstance(fill_value, Tensor) else fill_value.item(),
                              dtype=_dtype(dtype) or _default_dtype))


def zeros_like(x, dtype=None, name=None):
    return Tensor(_torch.zeros_like(x._t, dtype=_dtype(dtype)))


def ones_like(x, dtype=None, name=None):
    return Tensor(_torch.ones_like(x._t, dtype=_dtype(dtype)))


def full_like(x, fill_value, dtype=None, name=None):
    return Tensor(_torch.full_like(x._t, fill_value, dtype=_dtype(dtype)))


def arange(start=0, end=None, step=1, dtype=None, name=None):
    if end is None:
        start, end = 0, start
    start, end, step = _raw(start), _raw(end), _raw(step)
    dt = _dtype(dtype)
    if dt is None:
        ints = builtins.all(isinstance(v, (builtins.int, _np.integer)) for v in (start, end, step))
        dt = int64 if ints else _default_dtype
    return Tensor(_torch.arange(start, end, step, dtype=dt))


def randint(low=0, high=None, shape=(1,), dtype=None, name=None):
    if high is None:
        low, high = 0, low
    return Tensor(_torch.randint(_int(low), _int(high), tuple(_shape(shape)), dtype=_dtype(dtype) or int64))


def seed(s):
    _torch.manual_seed(s)


# --------------------------------------------------------------------------- manipulation
def shape(x):
    return Tensor(_torch.tensor(list(x._t.shape), dtype=int32))


def cast(x, dtype):
    return x.astype(dtype)


def transpose(x, perm, name=None):
    perm = [_int(p) for p in perm]
    if len(perm) != x._t.dim():
        raise ValueError(f"paddle shim: transpose perm {perm} for a {x._t.dim()}-D tensor")
    return Tensor(x._t.permute(*perm))


def reshape(x, shape, name=None):
    shp = _shape(shape)
    src = list(x._t.shape)
    out = []
    for i, s in enumerate(shp):
        if s == 0:  # paddle: 0 copies the corresponding input dimension
            s = src[i]
        out.append(s)
    return Tensor(x._t.reshape(out))


def unsqueeze(x, axis, name=None):
    if isinstance(axis, (list, tuple)):
        t = x._t
        for a in axis:  # paddle inserts the axes one after another, in the order given
            t = t.unsqueeze(_int(a))
        return Tensor(t)
    return Tensor(x._t.unsqueeze(_int(axis)))


def squeeze(x, axis=None, name=None):
    if axis is None:
        return Tensor(x._t.squeeze())
    if isinstance(axis, (list, tuple)):
        t = x._t
        for a in sorted((_int(a) % builtins.max(t.dim(), 1) for a in axis), reverse=True):
            if t.shape[a] == 1:
                t = t.squeeze(a)
        return Tensor(t)
    a = _int(axis)
    return Tensor(x._t.squeeze(a)) if x._t.shape[a] == 1 else Tensor(x._t)


def concat(x, axis=0, name=None):
    return Tensor(_torch.cat([_raw(v) for v in x], dim=_int(axis)))


def stack(x, axis=0, name=None):
    return Tensor(_torch.stack([_raw(v) for v in x], dim=_int(axis)))


def split(x, num_or_sections, axis=0, name=None):
    axis = _int(axis)
    n = x._t.shape[axis]
    if isinstance(num_or_sections, (builtins.int, _np.integer)):
        k = builtins.int(num_or_sections)  # NUMBER of equal sections (torch.split(int) would be the section SIZE)
        if n % k != 0:
            raise ValueError(f"paddle shim: split dim {n} not divisible into {k} sections")
        sizes = [n // k] * k
    else:
        sizes = [_int(s) for s in num_or_sections]
        if -1 in sizes:
            sizes[sizes.index(-1)] = n - (builtins.sum(sizes) + 1)
    return [Tensor(t) for t in _torch.split(x._t, sizes, dim=axis)]


def chunk(x, chunks, axis=0, name=None):
    return split(x, chunks, axis)


def expand(x, shape, name=None):
    return Tensor(x._t.expand(*_shape(shape)))


def broadcast_to(x, shape, name=None):
    return Tensor(x._t.broadcast_to(_shape(shape)))


def tile(x, repeat_times, name=None):
    return x.tile(repeat_times)


def flip(x, axis, name=None):
    axis = [axis] if isinstance(axis, (builtins.int, _np.integer)) else list(axis)
    return Tensor(_torch.flip(x._t, dims=axis))


def repeat_interleave(x, repeats, axis=None, name=None):
    return Tensor(_torch.repeat_interleave(x._t, _raw(repeats), dim=axis))


def masked_select(x, mask, name=None):
    return Tensor(_torch.masked_select(x._t, mask._t))


def where(condition, x=None, y=None, name=None):
    xx, yy = _raw(x), _raw(y)
    return Tensor(_torch.where(condition._t, xx, yy))


def tril(x, diagonal=0, name=None):
    return Tensor(_torch.tril(x._t, diagonal=_int(diagonal)))


def triu(x, diagonal=0, name=None):
    return Tensor(_torch.triu(x._t, diagonal=_int(diagonal)))


# --------------------------------------------------------------------------- math
def matmul(x, y, transpose_x=False, transpose_y=False, name=None):
    a, b = x._t, y._t
    if transpose_x:
        a = a.transpose(-1, -2)
    if transpose_y:
        b = b.transpose(-1, -2)
    return Tensor(_torch.matmul(a, b))


def add(x, y, name=None):
    return x + y


def multiply(x, y, name=None):
    return x * y


def exp(x, name=None):
    return Tensor(_torch.exp(x._t))


def log(x, name=None):
    return Tensor(_torch.log(x._t))


def sqrt(x, name=None):
    return Tensor(_torch.sqrt(x._t))


def sin(x, name=None):
    return Tensor(_torch.sin(x._t))


def cos(x, name=None):
    return Tensor(_torch.cos(x._t))


def tanh(x, name=None):
    return Tensor(_torch.tanh(x._t))


def abs(x, name=None):  # noqa: A001
    return Tensor(_torch.abs(x._t))


def maximum(x, y, name=None):
    return Tensor(_torch.maximum(x._t, y._t))


def minimum(x, y, name=None):
    return Tensor(_torch.minimum(x._t, y._t))


def clip(x, min=None, max=None, name=None):  # noqa: A002
    return Tensor(_torch.clamp(x._t, _raw(min), _raw(max)))


def _axes(axis):
    if axis is None:
        return None
    if isinstance(axis, (list, tuple)):
        return [_int(a) for a in axis]
    return _int(axis)


def sum(x, axis=None, dtype=None, keepdim=False, name=None):  # noqa: A001
    t = x._t
    if t.dtype in (_torch.bool, int32):  # paddle: bool / int32 sums come back as int64
        t = t.to(int64)
    ax = _axes(axis)
    r = t.sum() if ax is None else t.sum(dim=ax, keepdim=keepdim)
    if dtype is not None:
        r = r.to(_dtype(dtype))
    return Tensor(r)


def mean(x, axis=None, keepdim=False, name=None):
    ax = _axes(axis)
    return Tensor(x._t.mean() if ax is None else x._t.mean(dim=ax, keepdim=keepdim))


def max(x, axis=None, keepdim=False, name=None):  # noqa: A001
    ax = _axes(axis)
    return Tensor(x._t.max() if ax is None else x._t.amax(dim=ax, keepdim=keepdim))


def min(x, axis=None, keepdim=False, name=None):  # noqa: A001
    ax = _axes(axis)
    return Tensor(x._t.min() if ax is None else x._t.amin(dim=ax, keepdim=keepdim))


def argmax(x, axis=None, keepdim=False, dtype="int64", name=None):
    # first maximal index, like paddle / numpy
    if axis is None:
        r = _torch.argmax(x._t)
    else:
        r = _torch.argmax(x._t, dim=_int(axis), keepdim=keepdim)
    return Tensor(r.to(_dtype(dtype)))


def equal(x, y, name=None):
    return x == y


def logical_and(x, y, name=None):
    return Tensor(_torch.logical_and(x._t, _raw(y)))


def logical_or(x, y, name=None):
    return Tensor(_torch.logical_or(x._t, _raw(y)))


def logical_not(x, name=None):
    return Tensor(_torch.logical_not(x._t))


def allclose(x, y, rtol=1e-5, atol=1e-8, equal_nan=False, name=None):
    return Tensor(_torch.tensor(_torch.allclose(x._t, y._t, rtol=rtol, atol=atol, equal_nan=equal_nan)))


# --------------------------------------------------------------------------- framework bits
class ParamAttr:
    def __init__(self, name=None, initializer=None, learning_rate=1.0, regularizer=None, trainable=True,
                 do_model_average=True, need_clip=True):
        self.name = name
        self.initializer = initializer
        self.trainable = trainable


class no_grad:
    """Context manager and decorator (``@paddle.no_grad()``)."""

    def __init__(self, func=None):
        self._g = _torch.no_grad()

    def __enter__(self):
        return self._g.__enter__()

    def __exit__(self, *a):
        return self._g.__exit__(*a)

    def __call__(self, fn):
        def wrapped(*args, **kw):
            with _torch.no_grad():
                return fn(*args, **kw)
        wrapped.__name__ = getattr(fn, "__name__", "wrapped")
        return wrapped


def set_device(device):
    return device


def get_device():
    return "cpu"


def is_compiled_with_cuda():
    return False


def set_grad_enabled(mode):
    return _torch.set_grad_enabled(mode)


def save(obj, path, protocol=4, **kw):
    """``paddle.save(state_dict, path)``: a pickle of ``{name: ndarray}`` (what load_state_dict of the
    product reads).  Only state dicts are supported."""
    import pickle
    out = {k: (v.numpy() if isinstance(v, Tensor) else v) for k, v in obj.items()}
    with open(path, "wb") as f:
        pickle.dump(out, f, protocol=protocol)


def load(path, **kw):
    import pickle
    with open(path, "rb") as f:
        return pickle.load(f)


from . import nn  # noqa: E402,F401
from . import io  # noqa: E402,F401
from . import jit  # noqa: E402,F401
from . import static  # noqa: E402,F401
