"""Test-side WRITER of Paddle's inference-model files, following the layouts cited in
ppasr_amd/utils/paddle_inference.py (ProgramDesc protobuf, save_combine LoDTensor stream, .pdiparams.info pickle).
PaddlePaddle is not installable here, so reader and writer are both written from the documented format; the writer
deliberately uses the NON-packed encoding for `dims` and adds feed / fetch variables and unknown fields, as Paddle does."""
import pickle
import struct

import numpy as np

_TYPE_OF_NP = {np.dtype(np.bool_): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2, np.dtype(np.int64): 3,
               np.dtype(np.float16): 4, np.dtype(np.float32): 5, np.dtype(np.float64): 6}


def _varint(v):
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _key(fno, wt):
    return _varint((fno << 3) | wt)


def _ld(fno, payload):
    return _key(fno, 2) + _varint(len(payload)) + payload


def _vi(fno, v):
    return _key(fno, 0) + _varint(v)


def tensor_desc(dtype, dims):
    out = _vi(1, _TYPE_OF_NP[np.dtype(dtype)])
    for d in dims:
        out += _vi(2, d)  # proto2 repeated int64, not packed
    return out


def var_desc(name, dtype=None, dims=None, persistable=False, vtype=7):
    vt = _vi(1, vtype)
    if vtype == 7:
        vt += _ld(3, _ld(1, tensor_desc(dtype, dims)) + _vi(2, 0))
    out = _ld(1, name.encode()) + _ld(2, vt) + _vi(3, 1 if persistable else 0)
    out += _vi(5, 1 if persistable else 0)  # is_parameter: a field the reader does not use
    return out


def program_desc(persistables, extra_vars=()):
    """persistables: {program var name: ndarray}; the block lists the variables in an arbitrary (insertion) order."""
    block = _vi(1, 0) + _vi(2, -1)
    block += _ld(3, var_desc("feed", persistable=True, vtype=9))
    block += _ld(3, var_desc("fetch", persistable=True, vtype=10))
    for name, dtype, dims in extra_vars:
        block += _ld(3, var_desc(name, dtype, dims, persistable=False))
    for name, arr in persistables.items():
        block += _ld(3, var_desc(name, arr.dtype, arr.shape, persistable=True))
    block += _ld(4, _ld(3, b"feed"))  # an OpDesc (type = "feed"): skipped by the reader
    version = _ld(4, _vi(1, 0))
    return _ld(1, block) + version


def lod_tensor(arr):
    arr = np.ascontiguousarray(arr)
    desc = tensor_desc(arr.dtype, arr.shape)
    return (struct.pack("<I", 0) + struct.pack("<Q", 0) + struct.pack("<I", 0) + struct.pack("<i", len(desc)) + desc
            + arr.tobytes())


def write_inference_model(prefix, state_dict, constants=None):
    """state_dict {structured name: ndarray} -> prefix.pdmodel / .pdiparams / .pdiparams.info with Paddle-style program
    variable names (linear_3.w_0 ...).  Returns {program name: structured name}."""
    names = {}
    counters = {}
    for s, arr in state_dict.items():
        kind = "linear" if arr.ndim == 2 else ("conv" if arr.ndim >= 3 else "layer_norm")
        i = counters.get(kind, 0)
        counters[kind] = i + 1
        names[f"{kind}_{i}.{'w' if s.endswith('weight') else 'b'}_0"] = s
    prog = {p: np.asarray(state_dict[s]) for p, s in names.items()}
    for k, arr in (constants or {}).items():
        prog[k] = np.asarray(arr)
    with open(prefix + ".pdmodel", "wb") as f:
        f.write(program_desc(prog, extra_vars=[("speech", np.float32, (-1, -1, 80))]))
    with open(prefix + ".pdiparams", "wb") as f:
        for p in sorted(prog):  # save_combine order
            f.write(lod_tensor(prog[p]))
    info = {p: {"structured_name": s, "stop_gradient": False, "trainable": True} for p, s in names.items()}
    for k in (constants or {}):
        info[k] = {"stop_gradient": True}
    with open(prefix + ".pdiparams.info", "wb") as f:
        pickle.dump(info, f, protocol=2)
    return names
