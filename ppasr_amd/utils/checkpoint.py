"""Load a Paddle-layout parameter dict {name: float32 ndarray}.

* ``model.pdparams`` -- ``paddle.save(model.state_dict(), path)`` (trainer.py:302-328): a pickle of
  ``{structured name: ndarray}`` plus the bookkeeping entry ``StructuredToParameterName@@`` (a dict, skipped); some
  Paddle versions pickle ``(name, ndarray)`` tuples;
* ``.npz`` -- the same dict written with numpy;
* ``inference.pdmodel`` + ``inference.pdiparams`` (+ ``.info``) -- exported inference models (trainer.py:675-682),
  read by ``ppasr_amd/utils/paddle_inference.py``.

Names / layouts are consumed as they are by ``ppasr_create``.  Which names a real checkpoint carries was checked by
loading the synthetic dicts BY NAME into the reference's own model classes (tests/golden/make_ref_goldens.py):
besides the encoder + CTC head the reference's ``state_dict()`` holds the attention decoder (``decoder.*``, training
only), the unused ``concat_linear`` of the Efficient-Conformer's stride layer and, for ``nn.LSTM`` / ``nn.GRU``, every
parameter twice (``rnn.N.weight_ih_l0`` and ``rnn.N.0.cell.weight_ih``); extra names are ignored by ``ppasr_create``
and ``normalize_state_dict`` fills the ``*_l0`` names from the cell aliases when only those are present."""
import os
import pickle
import re

import numpy as np

__all__ = ["load_state_dict", "save_state_dict", "find_state_dict", "normalize_state_dict"]

_CELL = re.compile(r"^(.*\.rnn\.\d+)\.0\.(cell|cell_fw|cell_bw)\.(weight_ih|weight_hh|bias_ih|bias_hh)$")


def normalize_state_dict(sd):
    """RNN parameters under their cell names (``encoder.rnn.N.0.cell[_fw|_bw].weight_ih``) -> the ``*_l0[_reverse]``
    names the kernels look up (both forms exist in a real ``state_dict()``; a hand-made dict may carry only one)."""
    out = dict(sd)
    for k, v in sd.items():
        m = _CELL.match(k)
        if m:
            name = f"{m.group(1)}.{m.group(3)}_l0" + ("_reverse" if m.group(2) == "cell_bw" else "")
            out.setdefault(name, v)
    return out


def load_state_dict(path):
    if os.path.isdir(path):
        return load_state_dict(find_state_dict(path))
    if path.endswith(".pdmodel") or path.endswith(".pdiparams"):
        from ppasr_amd.utils.paddle_inference import load_inference_model
        stem = path.rsplit(".", 1)[0]
        info = stem + ".pdiparams.info"
        sd, _ = load_inference_model(stem + ".pdmodel", stem + ".pdiparams", info if os.path.exists(info) else None)
        return normalize_state_dict(sd)
    if path.endswith(".npz"):
        with np.load(path) as z:
            return normalize_state_dict({k: np.asarray(z[k], np.float32) for k in z.files})
    with open(path, "rb") as f:
        obj = pickle.load(f, encoding="latin1")
    out = {}
    for k, v in obj.items():
        if isinstance(v, np.ndarray):
            out[k] = v.astype(np.float32)
        elif isinstance(v, (tuple, list)) and len(v) == 2 and isinstance(v[1], np.ndarray):
            out[k] = v[1].astype(np.float32)  # some Paddle versions pickle (name, ndarray)
    if not out:
        raise ValueError(f"{path}: no ndarray entries (not a Paddle state dict?)")
    return normalize_state_dict(out)


def save_state_dict(sd, path):
    np.savez(path, **sd)


def find_state_dict(model_dir):
    for name in ("model.npz", "model.pdparams", "model_state.npz", "inference.pdmodel"):
        p = os.path.join(model_dir, name)
        if os.path.exists(p):
            return p
    raise Exception(f"no model.npz / model.pdparams / inference.pdmodel under {model_dir}")
