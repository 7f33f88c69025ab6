"""ORACLE (test infrastructure) -- ``paddle.inference`` stand-in, just enough for the reference's
``InferencePredictor`` (ppasr/infer_utils/inference_predictor.py:10-220) to run UNMODIFIED: named input handles
(``reshape`` / ``copy_from_cpu``), ``run()``, output handles (``copy_to_cpu``).

There is no ProgramDesc here.  ``run()`` calls the dygraph function that the reference's own ``model.export()``
(``paddle.jit.to_static`` over ``get_encoder_out`` / ``get_encoder_out_chunk``, e.g. conformer/model.py:187-206) hands to
``paddle.jit.save`` in ``trainer.py:675-682`` -- the exported graph computes what that method computes.  The test
generator registers that function under the model directory it passes to ``PPASRPredictor``::

    paddle.inference.register(model_dir, static_fn, input_names, dtypes)

Input order = the ``input_spec`` order of ``export()``.  The 1-element int32 inputs of the streaming graphs (``offset``,
``required_cache_size``: "int, but need be tensor", conformer/model.py:193-194) are handed to the dygraph method as the
Python ints its signature declares."""
import os

import numpy as _np

import paddle as _paddle

_REGISTRY = {}


def register(model_dir, fn, input_names, n_outputs):
    _REGISTRY[os.path.abspath(model_dir)] = (fn, list(input_names), int(n_outputs))


class PrecisionType:
    Float32 = 0
    Half = 1
    Int8 = 2


class Config:
    def __init__(self, model_path=None, params_path=None):
        self.model_dir = os.path.abspath(os.path.dirname(model_path)) if model_path else None

    def __getattr__(self, name):  # enable_use_gpu, disable_gpu, switch_ir_optim, ...: accepted and ignored
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: None


class _Handle:
    def __init__(self):
        self.value = None
        self._shape = None

    def reshape(self, shape):
        self._shape = list(shape)

    def copy_from_cpu(self, arr):
        a = _np.ascontiguousarray(arr)
        if self._shape is not None:
            assert list(a.shape) == self._shape, (a.shape, self._shape)
        self.value = a

    def copy_to_cpu(self):
        return self.value


class _Predictor:
    def __init__(self, fn, input_names, n_outputs):
        self._fn, self._names = fn, input_names
        self._in = {n: _Handle() for n in input_names}
        self._out_names = [f"output_{i}" for i in range(n_outputs)]
        self._out = {n: _Handle() for n in self._out_names}

    def get_input_handle(self, name):
        return self._in[name]

    def get_input_names(self):
        return list(self._names)

    def get_output_names(self):
        return list(self._out_names)

    def get_output_handle(self, name):
        return self._out[name]

    def run(self):
        args = []
        for n in self._names:
            v = self._in[n].value
            if n in ("offset", "required_cache_size"):
                assert v.shape == (1,) and v.dtype.kind == "i", (n, v.shape, v.dtype)
                args.append(int(v[0]))
            else:
                args.append(_paddle.to_tensor(v))
        with _paddle.no_grad():
            outs = self._fn(*args)
        if not isinstance(outs, (tuple, list)):
            outs = (outs,)
        for n, o in zip(self._out_names, outs):
            self._out[n].value = None if o is None else o.numpy()
        return True


def create_predictor(config):
    if config.model_dir not in _REGISTRY:
        raise RuntimeError(f"paddle shim: no function registered for {config.model_dir} (paddle.inference.register)")
    return _Predictor(*_REGISTRY[config.model_dir])
