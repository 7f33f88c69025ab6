"""ctypes binding of libppasr_hip.so (C-ABI in include/ppasr_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this
module raises.  The product path never routes through ``oracle/`` or any CPU
implementation.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPASR_HIP_LIB") or os.path.join(_HERE, "libppasr_hip.so")  # override: kernel experiments

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)

# ppasr_status (include/ppasr_hip.h)
PPASR_OK, PPASR_EINVAL, PPASR_EHIP, PPASR_EUNSUPPORTED, PPASR_EMISSING, PPASR_ENOSPACE = range(6)

PPASR_MODEL_CONFORMER = 0
PPASR_MODEL_EFFICIENT_CONFORMER = 1
PPASR_MODEL_SQUEEZEFORMER = 2
PPASR_MODEL_DEEPSPEECH2 = 3
N_KERNEL_CLASSES = 10
# ppasr_model_desc::options (include/ppasr_hip.h)
PPASR_OPT_POST_NORM, PPASR_OPT_CONCAT_AFTER, PPASR_OPT_NO_MACARON, PPASR_OPT_NO_CNN, PPASR_OPT_ACT_SHIFT = 4, 8, 16, 32, 8
PPASR_OPT_SQ_NO_ADAPTIVE_SCALE = 4096
PPASR_OPT_SQ_PRE_NORM = 8192
PPASR_GEMM_F32 = 0
PPASR_GEMM_F16X3 = 1
PPASR_GEMM_COVERS_LAYERS, PPASR_GEMM_COVERS_FRONT, PPASR_GEMM_COVERS_HEAD = 1, 2, 4
KPROF_NAME_LEN = 160


class WeightBlob(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data_host", ctypes.c_void_p), ("ndim", ctypes.c_int),
                ("shape", ctypes.c_int64 * 4)]


class ModelDesc(ctypes.Structure):
    _fields_ = [("model_type", ctypes.c_int), ("input_dim", ctypes.c_int), ("vocab_size", ctypes.c_int),
                ("output_size", ctypes.c_int), ("attention_heads", ctypes.c_int), ("linear_units", ctypes.c_int),
                ("num_blocks", ctypes.c_int), ("cnn_module_kernel", ctypes.c_int), ("causal", ctypes.c_int),
                ("max_len", ctypes.c_int), ("reduce_idx", ctypes.c_int), ("recover_idx", ctypes.c_int),
                ("stride_layer_idx", ctypes.c_int), ("group_layer_mask", ctypes.c_int), ("group_size", ctypes.c_int),
                ("use_gru", ctypes.c_int), ("input_layer", ctypes.c_int), ("options", ctypes.c_int),
                ("stride_layer_mask", ctypes.c_int)]


# every symbol include/ppasr_hip.h declares: (name, restype, argtypes)
_vp = ctypes.c_void_p
SYMBOLS = [
    ("ppasr_last_error", ctypes.c_char_p, []),
    ("ppasr_version", ctypes.c_char_p, []),
    ("ppasr_create", ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.POINTER(WeightBlob), ctypes.c_int,
                                    ctypes.POINTER(_vp)]),
    ("ppasr_destroy", ctypes.c_int, [_vp]),
    ("ppasr_out_frames", ctypes.c_int, [_vp, ctypes.c_int]),
    ("ppasr_workspace_bytes", ctypes.c_size_t, [_vp, ctypes.c_int, ctypes.c_int]),
    ("ppasr_encode", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp,
                                    ctypes.c_size_t, _vp]),
    ("ppasr_set_debug_taps", ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    ("ppasr_set_skip_padding", ctypes.c_int, [_vp, ctypes.c_int]),
    ("ppasr_set_row_block", ctypes.c_int, [_vp, ctypes.c_int]),
    ("ppasr_set_lengths_hint", ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    ("ppasr_set_ffn_split", ctypes.c_int, [_vp, ctypes.c_int]),
    ("ppasr_set_front_fused", ctypes.c_int, [_vp, ctypes.c_int]),
    ("ppasr_set_gemm_mode", ctypes.c_int, [_vp, ctypes.c_int]),
    ("ppasr_gemm_coverage", ctypes.c_int, [_vp]),
    ("ppasr_set_gemm_guard", ctypes.c_int, [_vp, ctypes.c_int]),
    ("ppasr_gemm_guard_stats", ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]),
    ("ppasr_edit_distance", ctypes.c_longlong, [_vp, ctypes.c_int, _vp, ctypes.c_int]),
    ("ppasr_ctc_beam_state_bytes", ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    ("ppasr_ctc_beam_search", ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             _vp, _vp, _vp, _vp, ctypes.c_size_t, ctypes.c_int, _vp]),
    ("ppasr_stream_create", ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    ("ppasr_stream_destroy", ctypes.c_int, [_vp]),
    ("ppasr_stream_reset", ctypes.c_int, [_vp, _vp]),
    ("ppasr_stream_offset", ctypes.c_int, [_vp]),
    ("ppasr_stream_cache_frames", ctypes.c_int, [_vp]),
    ("ppasr_chunk_workspace_bytes", ctypes.c_size_t, [_vp, ctypes.c_int]),
    ("ppasr_encode_chunk", ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp,
                                          ctypes.POINTER(ctypes.c_int), _vp, ctypes.c_size_t, _vp]),
    ("ppasr_stream_export_cache", ctypes.c_int, [_vp, _vp, _vp, _vp]),
    ("ppasr_stream_import_cache", ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, ctypes.c_int, _vp]),
    ("ppasr_ds2_workspace_bytes", ctypes.c_size_t, [_vp, ctypes.c_int, ctypes.c_int]),
    ("ppasr_ds2_encode", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                        ctypes.c_size_t, _vp]),
    ("ppasr_debug_occupy_cus", ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp]),
    ("ppasr_profile_enable", ctypes.c_int, [_vp, ctypes.c_int]),
    ("ppasr_profile_read", ctypes.c_int, [_vp, c_f32p, c_i32p]),
    ("ppasr_kernel_class_name", ctypes.c_char_p, [ctypes.c_int]),
    ("ppasr_kprof_begin", ctypes.c_int, []),
    ("ppasr_kprof_end", ctypes.c_int, [ctypes.c_int, _vp, c_f32p, c_i32p, ctypes.POINTER(ctypes.c_int)]),
    ("ppasr_ctc_greedy", ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp,
                                        _vp, _vp, ctypes.c_size_t, _vp]),
    ("ppasr_ctc_collapse", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp,
                                          _vp]),
    ("ppasr_lm_create_arpa", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int,
                                            ctypes.POINTER(_vp)]),
    ("ppasr_ctc_beam_scratch_bytes", ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_double, ctypes.c_int]),
    ("ppasr_ctc_beam_search_ws", ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                _vp, _vp, _vp, _vp, ctypes.c_size_t, ctypes.c_int, _vp, ctypes.c_double,
                                                ctypes.c_double, _vp, ctypes.c_size_t, _vp]),
    ("ppasr_ctc_beam_status", ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    ("ppasr_ctc_beam_state_grow", ctypes.c_int, [_vp, ctypes.c_size_t, _vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, _vp]),
    ("ppasr_hyp_pack", ctypes.c_int, [_vp, ctypes.c_longlong, ctypes.c_int, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp,
                                      ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    ("ppasr_hyp_unpack", ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    ("ppasr_lm_create", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int,
                                       ctypes.POINTER(_vp)]),
    ("ppasr_lm_create_klm", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int,
                                           ctypes.POINTER(_vp)]),
    ("ppasr_lm_debug_load_host", ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int,
                                                ctypes.POINTER(_vp)]),
    ("ppasr_lm_debug_host_score", ctypes.c_double, [_vp, _vp]),
    ("ppasr_lm_format", ctypes.c_char_p, [_vp]),
    ("ppasr_lm_word_index", ctypes.c_int, [_vp, ctypes.c_int]),
    ("ppasr_lm_bos", ctypes.c_int, [_vp]),
    ("ppasr_lm_eos", ctypes.c_int, [_vp]),
    ("ppasr_lm_destroy", ctypes.c_int, [_vp]),
    ("ppasr_lm_order", ctypes.c_int, [_vp]),
    ("ppasr_lm_is_character_based", ctypes.c_int, [_vp]),
    ("ppasr_lm_dict_size", ctypes.c_longlong, [_vp]),
    ("ppasr_lm_space_id", ctypes.c_int, [_vp]),
    ("ppasr_lm_ngram_count", ctypes.c_longlong, [_vp]),
    ("ppasr_ctc_beam_search_lm", ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                _vp, _vp, _vp, _vp, ctypes.c_size_t, ctypes.c_int, _vp, ctypes.c_double,
                                                ctypes.c_double, _vp]),
    ("ppasr_stream_group_create", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    ("ppasr_stream_group_destroy", ctypes.c_int, [_vp]),
    ("ppasr_stream_group_reset", ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    ("ppasr_stream_group_offset", ctypes.c_int, [_vp, ctypes.c_int]),
    ("ppasr_group_chunk_workspace_bytes", ctypes.c_size_t, [_vp, ctypes.c_int, ctypes.c_int]),
    ("ppasr_encode_chunk_group", ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.c_int, _vp, ctypes.c_int, _vp,
                                                _vp, _vp, ctypes.POINTER(ctypes.c_int), _vp, ctypes.c_size_t, _vp]),
    ("ppasr_fbank_create", ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                          ctypes.POINTER(_vp)]),
    ("ppasr_fbank_destroy", ctypes.c_int, [_vp]),
    ("ppasr_fbank_frames", ctypes.c_int, [_vp, ctypes.c_int]),
    ("ppasr_fbank_workspace_bytes", ctypes.c_size_t, [_vp, ctypes.c_int]),
    ("ppasr_fbank_compute", ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp, _vp,
                                           ctypes.c_size_t, _vp]),
]

_lib = None


class PPASRHipError(RuntimeError):
    """`status`: the library's numeric ppasr_status (PPASR_E*) when the error came through check(), else None -- callers
    that tolerate one specific refusal (PPASR_EUNSUPPORTED) test the code, never the message text."""

    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


def load():
    """Load (once) and type the shared library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PPASRHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = load().ppasr_last_error()
        raise PPASRHipError(f"libppasr_hip status {status}: {msg.decode() if msg else ''}", status=int(status))


class kernel_profile:
    """``with kernel_profile() as kp: ...`` -- every kernel this thread launches through the library inside the block
    carries its own dispatch-attached HIP events (``ppasr_kprof_*``); afterwards ``kp.kernels`` is
    ``{kernel name: (total_ms, launches)}`` with the names rocprofv3 prints (parameter lists dropped)."""

    _stack = []  # active scopes of this process, innermost last (the library keeps ONE recording per thread: a nested scope
                 # drains it at its boundaries and credits what it drained to every scope that was open meanwhile)

    def __init__(self, max_entries=128):
        self.max_entries = max_entries
        self.kernels = {}

    @staticmethod
    def _drain():
        n = max([k.max_entries for k in kernel_profile._stack] + [512])
        names = ctypes.create_string_buffer(n * KPROF_NAME_LEN)
        ms = (ctypes.c_float * n)()
        cnt = (ctypes.c_int32 * n)()
        got = ctypes.c_int(0)
        check(load().ppasr_kprof_end(n, ctypes.addressof(names), ms, cnt, ctypes.byref(got)))
        for i in range(got.value):
            nm = names.raw[i * KPROF_NAME_LEN:(i + 1) * KPROF_NAME_LEN].split(b"\0", 1)[0].decode()
            for scope in kernel_profile._stack:
                t, c = scope.kernels.get(nm, (0.0, 0))
                scope.kernels[nm] = (t + float(ms[i]), c + int(cnt[i]))

    def __enter__(self):
        if kernel_profile._stack:
            kernel_profile._drain()  # what ran so far belongs to the outer scopes only
        kernel_profile._stack.append(self)
        check(load().ppasr_kprof_begin())
        return self

    def __exit__(self, *exc):
        kernel_profile._drain()
        kernel_profile._stack.pop()
        if kernel_profile._stack:
            check(load().ppasr_kprof_begin())
        return False
