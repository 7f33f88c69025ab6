"""CPU checks of the drop-in boundary: the C-ABI library loads (no GPU needed) and exports every
symbol include/ppasr_hip.h declares; argument validation fails loudly; no compute calls here."""
import ctypes
import os
import re

import pytest

from ppasr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "ppasr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ppasr_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ppasr_hip.h but not exported"
    bound = {s[0] for s in _lib.SYMBOLS}
    assert set(names) == bound, (set(names) ^ bound)


def test_dynamic_symbol_table_is_exactly_the_header():
    """A drop-in library exports its header and nothing else: `nm -D --defined-only` == the PPASR_API declarations
    (-fvisibility=hidden + csrc/exports.map; no mangled internals, kernel handles or libstdc++ instantiations)."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == _declared_symbols(), sorted(set(exported) ^ set(_declared_symbols()))
    kinds = {line.split()[-2] for line in out.splitlines() if line.strip()}
    assert kinds == {"T"}, kinds
    # every declaration carries the export attribute
    src = open(os.path.join(ROOT, "include", "ppasr_hip.h")).read()
    assert src.count("PPASR_API ") - 1 == len(exported)  # (-1: the #define itself)


def test_version_and_error_strings():
    lib = _lib.load()
    assert b"gfx950" in lib.ppasr_version()
    # NULL arguments are rejected before any HIP call
    rc = lib.ppasr_create(None, None, 0, None)
    assert rc == 1 and b"null" in lib.ppasr_last_error()
    assert lib.ppasr_workspace_bytes(None, 1, 100) == 0
    with pytest.raises(_lib.PPASRHipError):
        _lib.check(lib.ppasr_set_debug_taps(None, None, 0))


def test_unsupported_configs_are_refused_not_emulated():
    lib = _lib.load()
    h = ctypes.c_void_p()
    blob = (_lib.WeightBlob * 1)()
    for desc in (_lib.ModelDesc(99, 80, 100, 256, 4, 2048, 2, 15, 1, 5000, -1, -1, -1, 0, 0),  # unknown model family
                 _lib.ModelDesc(0, 80, 100, 384, 6, 2048, 2, 15, 1, 5000, -1, -1, -1, 0, 0),   # width not a multiple of 256
                 _lib.ModelDesc(0, 80, 100, 512, 4, 2048, 2, 15, 1, 5000, -1, -1, -1, 0, 0),   # d_k != 64
                 _lib.ModelDesc(1, 80, 100, 256, 4, 2048, 2, 15, 1, 5000, -1, -1, -1, 0, 0, 0, 0, 4),  # options: conformer only
                 _lib.ModelDesc(0, 80, 100, 256, 4, 2000, 2, 15, 1, 5000, -1, -1, -1, 0, 0)):  # ffn % 256
        rc = lib.ppasr_create(ctypes.byref(desc), blob, 1, ctypes.byref(h))
        assert rc == 3, lib.ppasr_last_error()


def test_model_wrapper_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    with pytest.raises(_lib.PPASRHipError):
        ConformerModel(80, 10, state_dict={"x": [0.0]})


def test_option_codes_match_the_header():
    """ppasr_model_desc::options: the PPASR_OPT_* / PPASR_ACT_* values of include/ppasr_hip.h are the ones the Python layer
    packs (ppasr_amd/_lib.py, model_utils/conformer/model.py), and every activation name of the reference's get_activation
    (utils/common.py:189-206) has a code."""
    import re
    from ppasr_amd.model_utils.conformer import model as cm
    text = open(os.path.join(ROOT, "include", "ppasr_hip.h")).read()
    enums = {k: int(v) for k, v in re.findall(r"\b(PPASR_(?:OPT|ACT)_[A-Z0-9_]+)\s*=\s*(\d+)", text)}
    assert enums["PPASR_OPT_POS_REL"] == cm._POS_CODES["rel_pos"] == 0
    assert enums["PPASR_OPT_POS_ABS"] == cm._POS_CODES["abs_pos"] and enums["PPASR_OPT_POS_NONE"] == cm._POS_CODES["no_pos"]
    assert (enums["PPASR_OPT_POST_NORM"], enums["PPASR_OPT_CONCAT_AFTER"], enums["PPASR_OPT_NO_MACARON"], enums["PPASR_OPT_NO_CNN"],
            enums["PPASR_OPT_ACT_SHIFT"]) == (_lib.PPASR_OPT_POST_NORM, _lib.PPASR_OPT_CONCAT_AFTER, _lib.PPASR_OPT_NO_MACARON,
                                              _lib.PPASR_OPT_NO_CNN, _lib.PPASR_OPT_ACT_SHIFT)
    for name, code in cm._ACT_CODES.items():
        assert enums["PPASR_ACT_" + name.upper()] == code, name
    assert set(cm._ACT_CODES) == {"hardshrink", "hardswish", "hardtanh", "tanh", "relu", "relu6", "leakyrelu", "selu", "swish",
                                  "gelu", "elu"}
    assert max(cm._ACT_CODES.values()) <= enums["PPASR_OPT_ACT_MASK"]
    assert enums["PPASR_OPT_SQ_NO_ADAPTIVE_SCALE"] == _lib.PPASR_OPT_SQ_NO_ADAPTIVE_SCALE
    assert enums["PPASR_OPT_SQ_PRE_NORM"] == _lib.PPASR_OPT_SQ_PRE_NORM
    # the descriptor the bindings build has the header's field count (19 ints: + stride_layer_mask in round 6) and order
    end = text.index("} ppasr_model_desc;")
    hdr = text[text.rindex("typedef struct", 0, end):end]
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    fields = re.findall(r"\bint\s+(\w+)\s*;", hdr)
    assert fields == [f[0] for f in _lib.ModelDesc._fields_]
    assert ctypes.sizeof(_lib.ModelDesc) == 19 * ctypes.sizeof(ctypes.c_int)
