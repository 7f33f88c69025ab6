"""Load a Paddle-layout parameter dict {name: float32 ndarray}: ``.npz`` or a pickled dict
(``paddle.save(model.state_dict(), 'model.pdparams')`` writes a pickle of numpy arrays,
trainer.py:302-328).  Names / layouts are consumed as is by ``ppasr_create``."""
import os
import pickle

import numpy as np

__all__ = ["load_state_dict", "save_state_dict", "find_state_dict"]


def load_state_dict(path):
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: np.asarray(z[k], np.float32) for k in z.files}
    with open(path, "rb") as f:
        obj = pickle.load(f, encoding="latin1")
    out = {}
    for k, v in obj.items():
        if isinstance(v, np.ndarray):
            out[k] = v.astype(np.float32)
        elif isinstance(v, (tuple, list)) and len(v) == 2 and isinstance(v[1], np.ndarray):
            out[k] = v[1].astype(np.float32)  # some Paddle versions pickle (name, ndarray)
    return out


def save_state_dict(sd, path):
    np.savez(path, **sd)


def find_state_dict(model_dir):
    for name in ("model.npz", "model.pdparams", "model_state.npz"):
        p = os.path.join(model_dir, name)
        if os.path.exists(p):
            return p
    raise Exception(f"no model.npz / model.pdparams under {model_dir}")
