"""CPU: the weight importers -- pickled ``model.pdparams`` in the forms Paddle writes, ``.npz``, and
exported inference models (``.pdmodel`` ProgramDesc + ``.pdiparams`` save_combine stream + ``.pdiparams.info``)."""
import os
import pickle

import numpy as np
import pytest

from paddle_format_writer import lod_tensor, program_desc, write_inference_model
from ppasr_amd.utils.checkpoint import load_state_dict, normalize_state_dict, save_state_dict
from ppasr_amd.utils.paddle_inference import load_inference_model, persistable_names, read_pdiparams, read_pdmodel_vars
from ppasr_amd.utils.synth import conformer_state_dict, deepspeech2_state_dict


def _same(a, b):
    assert set(a) == set(b)
    for k in a:
        assert a[k].dtype == np.float32 and np.array_equal(a[k], b[k]), k


def test_pdparams_pickle_forms_and_npz(tmp_path):
    sd = conformer_state_dict(vocab_size=31, num_blocks=1, seed=3)
    # (a) paddle.save(state_dict): ndarrays + the StructuredToParameterName@@ bookkeeping dict
    obj = dict(sd)
    obj["StructuredToParameterName@@"] = {k: f"param_{i}" for i, k in enumerate(sd)}
    p = tmp_path / "model.pdparams"
    with open(p, "wb") as f:
        pickle.dump(obj, f, protocol=2)
    _same(load_state_dict(str(p)), sd)
    # (b) (name, ndarray) tuples, float64 payloads
    with open(p, "wb") as f:
        pickle.dump({k: (f"param_{i}", v.astype(np.float64)) for i, (k, v) in enumerate(sd.items())}, f, protocol=4)
    _same(load_state_dict(str(p)), sd)
    # (c) npz round trip, and the directory form
    save_state_dict(sd, str(tmp_path / "model.npz"))
    _same(load_state_dict(str(tmp_path / "model.npz")), sd)
    _same(load_state_dict(str(tmp_path)), sd)
    with open(p, "wb") as f:
        pickle.dump({"epoch": 3}, f)
    with pytest.raises(ValueError):
        load_state_dict(str(p))


def test_rnn_cell_aliases_are_normalised():
    sd = deepspeech2_state_dict(vocab_size=29, num_rnn_layers=2, streaming=False, seed=5)
    aliased = {}
    for k, v in sd.items():
        if ".rnn." in k:
            layer, leaf = k.rsplit(".", 1)
            rev = leaf.endswith("_reverse")
            base = leaf.replace("_l0_reverse", "").replace("_l0", "")
            aliased[f"{layer}.0.{'cell_bw' if rev else 'cell_fw'}.{base}"] = v
        else:
            aliased[k] = v
    out = normalize_state_dict(aliased)
    for k in sd:
        assert np.array_equal(out[k], sd[k]), k
    uni = deepspeech2_state_dict(vocab_size=29, num_rnn_layers=1, streaming=True, seed=6)
    al = {k.replace("rnn.0.weight_ih_l0", "rnn.0.0.cell.weight_ih"): v for k, v in uni.items()}
    assert np.array_equal(normalize_state_dict(al)["encoder.rnn.0.weight_ih_l0"], uni["encoder.rnn.0.weight_ih_l0"])


def test_inference_model_round_trip(tmp_path):
    sd = conformer_state_dict(vocab_size=37, num_blocks=2, seed=11, perturb_norm=True)
    consts = {"eager_tmp_0": np.arange(12, dtype=np.float32).reshape(1, 3, 4), "eager_tmp_1": np.array([16.0], np.float32)}
    prefix = str(tmp_path / "inference")
    names = write_inference_model(prefix, sd, consts)
    vars_ = read_pdmodel_vars(prefix + ".pdmodel")
    by = {v["name"]: v for v in vars_}
    assert by["feed"]["persistable"] and by["feed"]["type"] == 9 and by["speech"]["shape"] == [-1, -1, 80]
    order = persistable_names(vars_)
    assert order == sorted(list(names) + list(consts)) and "feed" not in order and "fetch" not in order
    tensors = read_pdiparams(prefix + ".pdiparams")
    assert len(tensors) == len(order)
    got, extras = load_inference_model(str(tmp_path))
    _same(got, sd)
    assert set(extras) == set(consts) and np.array_equal(extras["eager_tmp_0"], consts["eager_tmp_0"])
    _same(load_state_dict(prefix + ".pdmodel"), sd)       # the generic entry point
    _same(load_state_dict(prefix + ".pdiparams"), sd)
    # dtypes other than fp32 and packed dims are read too
    with open(prefix + ".pdiparams", "wb") as f:
        f.write(lod_tensor(np.arange(6, dtype=np.int64).reshape(2, 3)))
        f.write(lod_tensor(np.float16([1.5, -2.0])))
    a, b = read_pdiparams(prefix + ".pdiparams")
    assert a.dtype == np.int64 and a.shape == (2, 3) and b.dtype == np.float16 and b.tolist() == [1.5, -2.0]


def test_inference_model_errors(tmp_path):
    sd = {"ctc.ctc_lo.weight": np.ones((4, 5), np.float32), "ctc.ctc_lo.bias": np.zeros(5, np.float32)}
    prefix = str(tmp_path / "inference")
    write_inference_model(prefix, sd)
    # a parameter file that does not belong to the program
    with open(prefix + ".pdiparams", "ab") as f:
        f.write(lod_tensor(np.zeros(3, np.float32)))
    with pytest.raises(ValueError):
        load_inference_model(str(tmp_path))
    write_inference_model(prefix, sd)
    os.remove(prefix + ".pdiparams.info")
    with pytest.raises(ValueError):
        load_inference_model(str(tmp_path))
    with open(prefix + ".pdiparams", "r+b") as f:
        f.truncate(40)
    with pytest.raises(Exception):
        read_pdiparams(prefix + ".pdiparams")
    with open(prefix + ".pdmodel", "wb") as f:
        f.write(program_desc({})[:0])
    with pytest.raises(ValueError):
        read_pdmodel_vars(prefix + ".pdmodel")
