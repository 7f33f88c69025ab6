"""Reader for Paddle *inference models* -- what ``PPASRTrainer.export`` writes with ``paddle.jit.save``
(ppasr/trainer.py:675-682) and ``InferencePredictor`` opens (ppasr/infer_utils/inference_predictor.py:41-77):

    inference.pdmodel          serialized ``ProgramDesc`` protobuf (paddle/fluid/framework/framework.proto)
    inference.pdiparams        the persistable variables, ``save_combine`` format: one serialized LoDTensor after the
                               other, in the order of the SORTED variable names (python/paddle/static/io.py
                               ``_serialize_persistables`` / fluid io ``save_vars``: ``for name in sorted(save_var_map)``)
    inference.pdiparams.info   pickle ``{variable name: {'structured_name': 'encoder.embed.conv.0.weight', ...}}``
                               (python/paddle/jit/api.py ``save``: ``extra_var_info``)

No protobuf runtime and no Paddle are needed: the few message types involved are decoded with a hand-rolled varint
reader.  PaddlePaddle cannot be installed in the build container, so the byte layouts below are written from the Paddle
2.5 sources as recalled and are exercised against ``tests/paddle_format_writer.py`` (a writer that follows the same
documented layout), not against a file produced by Paddle itself -- DESIGN.md lists this under "parity unpinned".

ProgramDesc { repeated BlockDesc blocks = 1; }           BlockDesc { idx = 1; parent_idx = 2; repeated VarDesc vars = 3; ... }
VarDesc { string name = 1; VarType type = 2; bool persistable = 3; ... }
VarType { Type type = 1; LoDTensorDesc lod_tensor = 3; ... }   LoDTensorDesc { TensorDesc tensor = 1; int32 lod_level = 2; }
TensorDesc { Type data_type = 1; repeated int64 dims = 2; }
VarType.Type: BOOL 0, INT16 1, INT32 2, INT64 3, FP16 4, FP32 5, FP64 6, LOD_TENSOR 7, FEED_MINIBATCH 9, FETCH_LIST 10,
              RAW 17, UINT8 20, INT8 21

LoDTensor stream (paddle/fluid/framework/lod_tensor.cc ``SerializeToStream`` + tensor_util.cc ``TensorToStream``):
    u32 version (0) | u64 lod_level | per level: u64 byte size + data | u32 version (0) | i32 desc size |
    TensorDesc protobuf | raw little-endian data
"""
import os
import pickle
import struct

import numpy as np

__all__ = ["read_pdmodel_vars", "read_pdiparams", "load_inference_model", "find_inference_model"]

_NP_OF_TYPE = {0: np.bool_, 1: np.int16, 2: np.int32, 3: np.int64, 4: np.float16, 5: np.float32, 6: np.float64,
               20: np.uint8, 21: np.int8}
_LOD_TENSOR, _FEED, _FETCH, _RAW = 7, 9, 10, 17


# ---- minimal protobuf wire-format reader ----------------------------------------------------------------------------
def _varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 70:
            raise ValueError("protobuf: varint too long")


def _fields(buf):
    """Yield (field number, wire type, value) of one message; value is an int (varint / fixed) or a bytes slice."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            if len(v) != ln:
                raise ValueError("protobuf: truncated field")
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"protobuf: unsupported wire type {wt}")
        yield fno, wt, v


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _tensor_desc(buf):
    dtype, dims = None, []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dtype = v
        elif fno == 2:
            if wt == 2:  # packed
                p = 0
                while p < len(v):
                    d, p = _varint(v, p)
                    dims.append(_signed(d))
            else:
                dims.append(_signed(v))
    return dtype, dims


def read_pdmodel_vars(path):
    """-> list of dicts {name, persistable, type, dtype, shape} for block 0 of the ProgramDesc."""
    with open(path, "rb") as f:
        buf = f.read()
    out = []
    block_no = 0
    for fno, wt, blk in _fields(buf):
        if fno != 1 or wt != 2:
            continue
        if block_no == 0:
            for f2, w2, var in _fields(blk):
                if f2 != 3 or w2 != 2:
                    continue
                rec = {"name": None, "persistable": False, "type": None, "dtype": None, "shape": None}
                for f3, w3, v in _fields(var):
                    if f3 == 1:
                        rec["name"] = v.decode("utf-8")
                    elif f3 == 3:
                        rec["persistable"] = bool(v)
                    elif f3 == 2:
                        for f4, w4, t in _fields(v):
                            if f4 == 1:
                                rec["type"] = t
                            elif f4 == 3:  # LoDTensorDesc
                                for f5, w5, td in _fields(t):
                                    if f5 == 1:
                                        rec["dtype"], rec["shape"] = _tensor_desc(td)
                out.append(rec)
        block_no += 1
    if block_no == 0:
        raise ValueError(f"{path}: no BlockDesc found (not a ProgramDesc?)")
    return out


def persistable_names(vars_):
    """Names ``save_combine`` wrote, in file order: persistable, not feed / fetch / RAW, sorted by name."""
    return sorted(v["name"] for v in vars_ if v["persistable"] and v["type"] not in (_FEED, _FETCH, _RAW))


def read_pdiparams(path):
    """-> list of numpy arrays in file order."""
    with open(path, "rb") as f:
        buf = f.read()
    pos, out = 0, []
    while pos < len(buf):
        (ver,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        if ver != 0:
            raise ValueError(f"{path}: LoDTensor version {ver} at byte {pos - 4}")
        (lod_levels,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        for _ in range(lod_levels):
            (nbytes,) = struct.unpack_from("<Q", buf, pos)
            pos += 8 + nbytes
        (tver,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        if tver != 0:
            raise ValueError(f"{path}: tensor version {tver}")
        (dsize,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        dtype, dims = _tensor_desc(buf[pos:pos + dsize])
        pos += dsize
        if dtype not in _NP_OF_TYPE:
            raise ValueError(f"{path}: unsupported tensor data type {dtype}")
        npdt = np.dtype(_NP_OF_TYPE[dtype])
        count = int(np.prod(dims)) if dims else 1
        nbytes = count * npdt.itemsize
        if pos + nbytes > len(buf):
            raise ValueError(f"{path}: truncated tensor data")
        out.append(np.frombuffer(buf, dtype=npdt, count=count, offset=pos).reshape(dims).copy())
        pos += nbytes
    return out


def find_inference_model(model_dir):
    """-> (pdmodel, pdiparams, info or None) for the file names PPASR exports (inference_predictor.py:41-42)."""
    for stem in ("inference", "model"):
        m, p = os.path.join(model_dir, stem + ".pdmodel"), os.path.join(model_dir, stem + ".pdiparams")
        if os.path.exists(m) and os.path.exists(p):
            info = p + ".info"
            return m, p, info if os.path.exists(info) else None
    return None


_warned = False


def load_inference_model(model_dir_or_pdmodel, pdiparams=None, info=None):
    """-> ({structured name: float32 ndarray}, extras) ready for the ``state_dict=`` argument of the ppasr_amd models.

    Variables are matched to their structured (``state_dict``) names through ``.pdiparams.info``; variables without a
    structured name (constants the dygraph-to-static pass captured, e.g. the positional table) come back in ``extras``
    under their program names.  Shapes in the parameter file are checked against the ProgramDesc."""
    global _warned
    if not _warned:  # (once per process; ADVICE r02: the byte layouts were written from the Paddle sources as recalled)
        _warned = True
        import warnings
        warnings.warn("exported Paddle inference model (.pdmodel / .pdiparams) read by an UNVERIFIED reader: it was never "
                      "checked against a file written by Paddle itself (none is reachable offline), only against "
                      "tests/paddle_format_writer.py; tensor counts and shapes are cross-checked between the two files, "
                      "values are taken as found", RuntimeWarning, stacklevel=2)
    if pdiparams is None:
        found = find_inference_model(model_dir_or_pdmodel)
        if found is None:
            raise FileNotFoundError(f"no inference.pdmodel / inference.pdiparams under {model_dir_or_pdmodel}")
        pdmodel, pdiparams, info = found
    else:
        pdmodel = model_dir_or_pdmodel
    vars_ = read_pdmodel_vars(pdmodel)
    names = persistable_names(vars_)
    tensors = read_pdiparams(pdiparams)
    if len(names) != len(tensors):
        raise ValueError(f"{pdiparams}: {len(tensors)} tensors for {len(names)} persistable variables of {pdmodel}")
    by_name = {v["name"]: v for v in vars_}
    structured = {}
    if info is not None:
        with open(info, "rb") as f:
            meta = pickle.load(f, encoding="latin1")
        structured = {k: v.get("structured_name") for k, v in meta.items() if isinstance(v, dict)}
    sd, extras = {}, {}
    for name, arr in zip(names, tensors):
        want = by_name[name]["shape"]
        if want is not None and [d for d in want] != list(arr.shape) and not any(d < 0 for d in want):
            raise ValueError(f"{name}: shape {list(arr.shape)} in {pdiparams} vs {want} in {pdmodel}")
        s = structured.get(name)
        if s:
            sd[s] = arr.astype(np.float32) if arr.dtype.kind == "f" else arr
        else:
            extras[name] = arr
    if not sd:
        raise ValueError("no variable has a structured name: inference.pdiparams.info is needed to map program variable "
                         "names (linear_0.w_0, ...) to state_dict names")
    return sd, extras
