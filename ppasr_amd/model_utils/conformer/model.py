"""Host-side mirror of ``ppasr/model_utils/conformer/model.py`` (``ConformerModel``),
inference surface only: ``get_encoder_out`` (:148-162) and ``get_encoder_out_chunk``
(:164-184), backed by the HIP kernels through the C-ABI (``include/ppasr_hip.h``).

torch is used for device memory and streams only; all compute is in libppasr_hip.so.
"""
import ctypes
import json
import math

import numpy as np
import torch

from ppasr_amd import _lib

__all__ = ["ConformerModel"]

# ppasr_model_desc::options codes (include/ppasr_hip.h)
_POS_CODES = {"rel_pos": 0, "abs_pos": 1, "no_pos": 2}
_ACT_CODES = {"swish": 0, "relu": 1, "gelu": 2, "tanh": 3, "hardtanh": 4, "relu6": 5, "leakyrelu": 6, "selu": 7, "elu": 8,
              "hardswish": 9, "hardshrink": 10}


def _pe_table(d_model, max_len):
    # PositionalEncoding.__init__  (conformer/embedding.py:38-53), fp32 like the reference
    pe = torch.zeros(max_len, d_model, dtype=torch.float32)
    position = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.numpy()


class ConformerModel:
    """Drop-in for the inference half of the reference ``ConformerModel``.

    Args mirror ``conformer/model.py:17-29``; the training-only arguments
    (decoder_conf, ctc_weight, ...) are accepted and ignored.  ``state_dict`` is a
    Paddle-layout ``{name: np.ndarray}`` (what ``paddle.load('model.pdparams')`` holds).
    """

    def __init__(self, input_dim, vocab_size, mean_istd_path=None, streaming=True, encoder_conf=None,
                 decoder_conf=None, ctc_weight=0.5, state_dict=None, device="cuda:0", **_ignored):
        if state_dict is None:
            raise ValueError("state_dict (Paddle-layout parameter dict) is required")
        if not torch.cuda.is_available():
            raise _lib.PPASRHipError("no HIP device visible: ppasr_amd has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.input_dim = input_dim
        self.vocab_size = vocab_size
        self.streaming = streaming
        conf = dict(encoder_conf or {})
        self.output_size = int(conf.get("output_size", 256))
        self.attention_heads = int(conf.get("attention_heads", 4))
        self.linear_units = int(conf.get("linear_units", 2048))
        self.num_blocks = int(conf.get("num_blocks", 6))
        self.cnn_module_kernel = int(conf.get("cnn_module_kernel", 15))
        self.max_len = int(conf.get("max_len", 5000))
        # The remaining ConformerEncoder constructor arguments (conformer/encoder.py:38-48).  The shipped values run on
        # the fused kernels; anything else (like output_size != 256) selects the library's general layer route.
        pos = conf.get("pos_enc_layer_type", "rel_pos")
        act = conf.get("activation_type", "swish")
        if pos not in _POS_CODES:
            raise ValueError("unknown pos_enc_layer: " + str(pos))  # encoder.py:102
        if act not in _ACT_CODES:
            raise KeyError(act)  # get_activation (utils/common.py:206)
        self.options = (_POS_CODES[pos] | (0 if conf.get("normalize_before", True) else _lib.PPASR_OPT_POST_NORM)
                        | (_lib.PPASR_OPT_CONCAT_AFTER if conf.get("concat_after", False) else 0)
                        | (0 if conf.get("macaron_style", True) else _lib.PPASR_OPT_NO_MACARON)
                        | (0 if conf.get("use_cnn_module", True) else _lib.PPASR_OPT_NO_CNN)
                        | (_ACT_CODES[act] << _lib.PPASR_OPT_ACT_SHIFT))
        self.use_cnn_module = bool(conf.get("use_cnn_module", True))
        # input_layer (encoder.py:104-113): LinearNoSubsampling, Conv2dSubsampling4, or the 6x / 8x variants
        il = conf.get("input_layer", "conv2d")
        if il not in ("linear", "conv2d", "conv2d6", "conv2d8"):
            raise ValueError("unknown input_layer: " + str(il))
        self.input_layer = il
        self.subsampling_rate = {"linear": 1, "conv2d": 4, "conv2d6": 6, "conv2d8": 8}[il]
        # cnn_module_norm (convolution.py:65-71): layer_norm, or batch_norm = nn.BatchNorm1D in eval mode, which the
        # library folds into a per-channel scale / shift; the checkpoint carries the running statistics then
        norm = conf.get("cnn_module_norm", "layer_norm")
        if norm not in ("layer_norm", "batch_norm"):
            raise ValueError(f"encoder_conf.cnn_module_norm={norm!r}")
        has_stats = any(k.endswith("conv_module.norm._mean") for k in state_dict)
        if self.use_cnn_module and has_stats != (norm == "batch_norm"):
            raise ValueError(f"encoder_conf.cnn_module_norm={norm!r} but the checkpoint "
                             f"{'has' if has_stats else 'lacks'} conv_module.norm._mean / _variance")
        sd = dict(state_dict)
        if mean_istd_path is not None:
            # FeatureNormalizer JSON {"mean": [...], "istd": [...]}  (data_utils/normalizer.py:35-41)
            with open(mean_istd_path, "r", encoding="utf-8") as f:
                js = json.load(f)
            sd["encoder.global_cmvn.mean"] = np.asarray(js["mean"], np.float32)
            sd["encoder.global_cmvn.istd"] = np.asarray(js["istd"], np.float32)
        sd["__pe_table__"] = _pe_table(self.output_size, self.max_len)
        keep = []  # keep the numpy buffers alive during ppasr_create
        blobs = (_lib.WeightBlob * len(sd))()
        for i, (name, arr) in enumerate(sd.items()):
            a = np.ascontiguousarray(arr, dtype=np.float32)
            if a.ndim > 4:
                raise ValueError(f"{name}: ndim > 4")
            keep.append(a)
            blobs[i].name = name.encode()
            blobs[i].data_host = a.ctypes.data
            blobs[i].ndim = a.ndim
            for j in range(a.ndim):
                blobs[i].shape[j] = a.shape[j]
        desc = _lib.ModelDesc(_lib.PPASR_MODEL_CONFORMER, input_dim, vocab_size, self.output_size,
                              self.attention_heads, self.linear_units, self.num_blocks, self.cnn_module_kernel,
                              1 if streaming else 0, self.max_len, -1, -1, -1, 0, 0, 0,
                              {"conv2d": 0, "linear": 1, "conv2d6": 6, "conv2d8": 8}[self.input_layer], self.options)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ppasr_create(ctypes.byref(desc), blobs, len(sd), ctypes.byref(handle)))
        self._h = handle
        self._ws = None
        self._taps = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self.lib.ppasr_destroy(h)
            self._h = None

    # ------------------------------------------------------------------
    def out_frames(self, T):
        return int(self.lib.ppasr_out_frames(self._h, int(T)))

    def _workspace(self, B, T):
        """One workspace per HIP stream: encodes issued on different streams (length buckets that do not fill the
        chip on their own) may overlap; the handle itself is read-only during an encode."""
        need = int(self.lib.ppasr_workspace_bytes(self._h, B, T))
        key = torch.cuda.current_stream(self.device).cuda_stream
        if not isinstance(self._ws, dict):
            self._ws = {}
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def _prep(self, speech, speech_lengths):
        speech = torch.as_tensor(speech, dtype=torch.float32).to(self.device).contiguous()
        lens = torch.as_tensor(speech_lengths, dtype=torch.int64).to(self.device).contiguous()
        assert speech.dim() == 3 and speech.shape[2] == self.input_dim and lens.shape[0] == speech.shape[0]
        return speech, lens

    def _encode(self, speech, lens, probs=None, logits=None, fa=None, fp=None):
        B, T, _ = speech.shape
        ws = self._workspace(B, T)
        ptr = lambda t: None if t is None else t.data_ptr()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.ppasr_encode(self._h, speech.data_ptr(), lens.data_ptr(), B, T, ptr(probs), ptr(logits),
                                         ptr(fa), ptr(fp), ws.data_ptr(), ws.numel(), stream))

    def get_encoder_out(self, speech, speech_lengths, return_logits=False):
        """-> ctc_probs [B, T', V] (device tensor).  conformer/model.py:148-162"""
        speech, lens = self._prep(speech, speech_lengths)
        B, T, _ = speech.shape
        Tp = self.out_frames(T)
        probs = torch.empty(B, Tp, self.vocab_size, dtype=torch.float32, device=self.device)
        logits = torch.empty_like(probs) if return_logits else None
        with torch.cuda.device(self.device):
            self._encode(speech, lens, probs=probs, logits=logits)
        return (probs, logits) if return_logits else probs

    def encode_greedy(self, speech, speech_lengths, trim_to_length=False, blank=0):
        """Fused path: features -> (tokens [B,T'] i32 (-1 padded), n_tokens [B] i32, score [B] f64), all on
        device; the [B,T',V] probability tensor is never materialised.  Equivalent to
        ``greedy_decoder_batch(get_encoder_out(...))`` (trainer.py:626,351); ``trim_to_length=False``
        reproduces the reference, which decodes all T' rows including PAD frames."""
        speech, lens = self._prep(speech, speech_lengths)
        B, T, _ = speech.shape
        Tp = self.out_frames(T)
        fa = torch.empty(B, Tp, dtype=torch.int32, device=self.device)
        fp = torch.empty(B, Tp, dtype=torch.float32, device=self.device)
        tokens = torch.empty(B, Tp, dtype=torch.int32, device=self.device)
        n_tokens = torch.empty(B, dtype=torch.int32, device=self.device)
        score = torch.empty(B, dtype=torch.float64, device=self.device)
        frame_lens = None
        if trim_to_length:
            frame_lens = self.valid_out_frames(lens, T)  # frame t valid iff mul * t < len (mul = 4, or 8 behind a stride layer)
        with torch.cuda.device(self.device):
            self._encode(speech, lens, fa=fa, fp=fp)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self.lib.ppasr_ctc_collapse(fa.data_ptr(), fp.data_ptr(),
                                                   None if frame_lens is None else frame_lens.data_ptr(), B, Tp,
                                                   blank, tokens.data_ptr(), n_tokens.data_ptr(), score.data_ptr(),
                                                   stream))
        return tokens, n_tokens, score

    def valid_out_frames(self, speech_lengths, T):
        """Output frames that are not padding, per utterance (int32 device tensor): frame t of the output is valid iff
        ``mul * t < len`` with ``mul`` = the model's total time reduction (the reference's mask slicing,
        subsampling.py:115; x2 behind the Efficient-Conformer's stride layer)."""
        lens = torch.as_tensor(speech_lengths, dtype=torch.int64).to(self.device)
        mul = getattr(self, "subsampling_rate", 4) * (2 ** len(getattr(self, "_stride_layers", ())))
        return torch.clamp((lens + mul - 1) // mul, min=0, max=self.out_frames(int(T))).to(torch.int32)

    def set_skip_padding(self, enable=True):
        """Ragged batches: compute, per utterance, only the rows its valid output frames depend on
        (``ppasr_set_skip_padding``).  Valid rows are bit-identical to the default mode; rows of the returned
        probabilities behind an utterance's last valid frame (``4 t >= len``; ``8 t`` for the Efficient-Conformer)
        are 0, so decode with ``frame_lens`` / ``trim_to_length=True``.  Off by default: the reference computes (and
        its batched decoders consume, trainer.py:347) every padded row."""
        self.skip_padding = bool(enable)
        _lib.check(self.lib.ppasr_set_skip_padding(self._h, 1 if enable else 0))

    def set_ffn_split(self, mode=-1):
        """Under-filled launches (``ppasr_set_ffn_split``): -1 = split the feed-forward modules' hidden dimension over
        2 / 4 / 8 workgroups per row block when a call has <= 128 row blocks (default), 0 = always the fused kernels,
        2 / 4 / 8 = always that many slices."""
        _lib.check(self.lib.ppasr_set_ffn_split(self._h, int(mode)))

    def set_front_fused(self, mode=-1):
        """Conv2dSubsampling4 as one launch (-1 / 1, default) or two (0): ``ppasr_set_front_fused``; same results bit for
        bit.  Two launches are the faster form when another stream's kernels (a pipelined beam search) share the GPU."""
        _lib.check(self.lib.ppasr_set_front_fused(self._h, int(mode)))
        self.front_fused = int(mode)  # (callers that switch it for one call restore this: parallel.RaggedPlan)

    def set_gemm_mode(self, mode="f32"):
        """Arithmetic of the feed-forward GEMMs (``ppasr_set_gemm_mode``): "f32" (default, exact fp32 products on
        ``v_mfma_f32_32x32x2_f32``) or "f16x3" (opt-in: two fp16 pieces per operand, three fp16 MFMAs per 16-wide k step,
        fp32 accumulation -- closer to float64 than fp32 arithmetic on one module, not bit-identical to "f32"; plain
        Conformer on the fused route, full 32-row launches)."""
        modes = {"f32": _lib.PPASR_GEMM_F32, "f16x3": _lib.PPASR_GEMM_F16X3}
        if mode not in modes:
            raise ValueError(f"gemm mode {mode!r}: 'f32' or 'f16x3'")
        _lib.check(self.lib.ppasr_set_gemm_mode(self._h, modes[mode]))

    def gemm_coverage(self):
        """Which parts the current GEMM mode switched (``ppasr_gemm_coverage``): a set out of {"layers", "front", "head"};
        empty in "f32"."""
        bits = int(self.lib.ppasr_gemm_coverage(self._h))
        names = ((_lib.PPASR_GEMM_COVERS_LAYERS, "layers"), (_lib.PPASR_GEMM_COVERS_FRONT, "front"),
                 (_lib.PPASR_GEMM_COVERS_HEAD, "head"))
        return {n for b, n in names if bits & b}

    def set_gemm_guard(self, enable=True):
        """Range guard of the "f16x3" mode (``ppasr_set_gemm_guard``).  On (default): a call whose GEMM inputs left the
        fp16 pieces' range (|activation| > 4 094) is run again on the fp32 kernels before ``ppasr_encode`` returns -- the
        call then synchronises its stream.  Off: calls stay asynchronous, out-of-range inputs are saturated (never Inf /
        NaN) and counted; poll ``gemm_guard_stats``.

        STREAMING IS EXCLUDED from the re-run: ``get_encoder_out_chunk`` / stream handles / session groups
        (``ppasr_encode_chunk``, ``ppasr_encode_chunk_group``) only saturate and count -- a chunk is never run again, and
        the saturated values are what enter the K / V and conv caches, i.e. they colour every later chunk of that session.
        Serving code that runs the mode on streams should snapshot ``gemm_guard_stats()`` around a chunk and, when the event
        count moved, reset the session (``reset_stream``) or replay it with ``set_gemm_mode("f32")``."""
        _lib.check(self.lib.ppasr_set_gemm_guard(self._h, 1 if enable else 0))

    def gemm_guard_stats(self):
        """-> (calls re-run on the fp32 kernels, saturation events seen) of this handle (``ppasr_gemm_guard_stats``)."""
        f, e = ctypes.c_longlong(0), ctypes.c_longlong(0)
        _lib.check(self.lib.ppasr_gemm_guard_stats(self._h, ctypes.byref(f), ctypes.byref(e)))
        return int(f.value), int(e.value)

    def set_row_block(self, rows=-1):
        """Block form of the layer kernels (``ppasr_set_row_block``): -1 = by grid size (16-row blocks for under-filled
        launches, else 32), 16 / 32 = always, 1032 (``PPASR_ROW_BLOCK_32_W16``) = 32 rows on 16 waves (optional form)."""
        _lib.check(self.lib.ppasr_set_row_block(self._h, int(rows)))

    def set_lengths_hint(self, lengths=None):
        """Host copy of the lengths of the batches the following calls encode (``ppasr_set_lengths_hint``; None: forget).
        Route selection for ragged batches only -- a wrong hint costs speed, never correctness."""
        if lengths is None:
            _lib.check(self.lib.ppasr_set_lengths_hint(self._h, None, 0))
            return
        arr = (ctypes.c_int64 * len(lengths))(*[int(v) for v in lengths])
        _lib.check(self.lib.ppasr_set_lengths_hint(self._h, arr, len(lengths)))

    def set_debug_taps(self, n_floats):
        """Allocate a tap buffer; layout in DESIGN.md (x0, then per layer x1,qkv,ctx,x2,g,x_out)."""
        self._taps = torch.zeros(n_floats, dtype=torch.float32, device=self.device) if n_floats else None
        _lib.check(self.lib.ppasr_set_debug_taps(self._h, None if self._taps is None else self._taps.data_ptr(),
                                                 n_floats))
        return self._taps

    def profile_kernels(self, enable=True):
        _lib.check(self.lib.ppasr_profile_enable(self._h, 1 if enable else 0))

    def read_kernel_profile(self):
        """-> {kernel class name: (total_ms, launches)} for the last profiled encode (synchronises)."""
        ms = (ctypes.c_float * _lib.N_KERNEL_CLASSES)()
        n = (ctypes.c_int32 * _lib.N_KERNEL_CLASSES)()
        _lib.check(self.lib.ppasr_profile_read(self._h, ms, n))
        return {self.lib.ppasr_kernel_class_name(i).decode(): (float(ms[i]), int(n[i]))
                for i in range(_lib.N_KERNEL_CLASSES)}

    # ---- streaming -----------------------------------------------------------------
    def new_stream(self):
        """Device-resident streaming state (attention K/V cache, conv cache, offset) for one session."""
        return ConformerStream(self)

    def get_encoder_out_chunk(self, speech, offset, required_cache_size, att_cache=None, cnn_cache=None):
        """Stateless signature of the reference (conformer/model.py:164-184):
        (speech [1,t,F], offset, required_cache_size, att_cache [L,h,t1,2dk], cnn_cache [L,1,d,k-1])
        -> (ctc_probs [1,c,V], new att_cache, new cnn_cache), device tensors in the reference layouts.
        The caches are imported into / exported from a scratch stream object; sessions that keep the
        state on the device should use ``new_stream()`` (what InferencePredictor does)."""
        if getattr(self, "_scratch_stream", None) is None:
            self._scratch_stream = self.new_stream()
        s = self._scratch_stream
        s.load_caches(att_cache, cnn_cache, int(offset))
        probs = s.encode_chunk(speech, int(required_cache_size))
        att, cnn = s.export_caches()
        return probs, att, cnn


class ConformerStream:
    def __init__(self, model):
        self.model = model
        self.lib = model.lib
        self._s = ctypes.c_void_p()
        with torch.cuda.device(model.device):
            _lib.check(self.lib.ppasr_stream_create(model._h, ctypes.byref(self._s)))
        self._ws = None

    def __del__(self):
        s = getattr(self, "_s", None)
        if s is not None and s.value:
            self.lib.ppasr_stream_destroy(s)
            self._s = None

    @property
    def offset(self):
        return int(self.lib.ppasr_stream_offset(self._s))

    @property
    def cache_frames(self):
        return int(self.lib.ppasr_stream_cache_frames(self._s))

    def reset(self):
        m = self.model
        with torch.cuda.device(m.device):
            _lib.check(self.lib.ppasr_stream_reset(self._s, torch.cuda.current_stream(m.device).cuda_stream))

    def encode_chunk(self, speech, required_cache_size=-1, want_probs=True, want_frames=False):
        """speech [1,t,F] -> ctc_probs [1,c,V] (device tensor); caches / offset advance on the device."""
        m = self.model
        x = torch.as_tensor(speech, dtype=torch.float32).to(m.device).contiguous()
        assert x.dim() == 3 and x.shape[0] == 1 and x.shape[2] == m.input_dim  # encoder.py:238
        T = int(x.shape[1])
        c = m.out_frames(T)
        need = int(self.lib.ppasr_chunk_workspace_bytes(m._h, T))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=m.device)
        probs = torch.empty(1, max(c, 0), m.vocab_size, dtype=torch.float32, device=m.device) if want_probs else None
        fa = torch.empty(1, max(c, 0), dtype=torch.int32, device=m.device) if want_frames else None
        fp = torch.empty(1, max(c, 0), dtype=torch.float32, device=m.device) if want_frames else None
        c_out = ctypes.c_int(0)
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(m.device):
            stream = torch.cuda.current_stream(m.device).cuda_stream
            _lib.check(self.lib.ppasr_encode_chunk(self._s, x.data_ptr(), T, int(required_cache_size), ptr(probs),
                                                   ptr(fa), ptr(fp), ctypes.byref(c_out), self._ws.data_ptr(),
                                                   self._ws.numel(), stream))
        if want_frames:
            return probs, fa, fp
        return probs

    def export_caches(self):
        """-> (att_cache [L,h,t,2dk], cnn_cache [L,1,d,k-1]) device tensors, reference layouts."""
        m = self.model
        t = self.cache_frames
        att = torch.empty(m.num_blocks, m.attention_heads, t, 2 * (m.output_size // m.attention_heads),
                          dtype=torch.float32, device=m.device)
        # (use_cnn_module = False: every layer returns zeros([0, 0, 0]) as its conv cache, encoder.py:405 -> [L, 0, 0, 0])
        if getattr(m, "use_cnn_module", True):
            cnn = torch.empty(m.num_blocks, 1, m.output_size, m.cnn_module_kernel - 1, dtype=torch.float32, device=m.device)
        else:
            cnn = torch.empty(m.num_blocks, 0, 0, 0, dtype=torch.float32, device=m.device)
        with torch.cuda.device(m.device):
            stream = torch.cuda.current_stream(m.device).cuda_stream
            _lib.check(self.lib.ppasr_stream_export_cache(self._s, att.data_ptr() if t > 0 else None,
                                                          cnn.data_ptr() if cnn.numel() else None,
                                                          stream))
        return att, cnn

    def load_caches(self, att_cache, cnn_cache, offset):
        m = self.model
        att = cnn = None
        t = 0
        if att_cache is not None and att_cache.numel() > 0:
            att = torch.as_tensor(att_cache, dtype=torch.float32).to(m.device).contiguous()
            t = int(att.shape[2])
        if cnn_cache is not None and cnn_cache.numel() > 0:
            cnn = torch.as_tensor(cnn_cache, dtype=torch.float32).to(m.device).contiguous()
        with torch.cuda.device(m.device):
            stream = torch.cuda.current_stream(m.device).cuda_stream
            _lib.check(self.lib.ppasr_stream_import_cache(self._s, None if att is None else att.data_ptr(), t,
                                                          None if cnn is None else cnn.data_ptr(), int(offset),
                                                          stream))
            torch.cuda.current_stream(m.device).synchronize()  # att / cnn temporaries must outlive the copy


class StreamHandleSet:
    """The interface of ``ConformerStreamGroup`` over per-session stream handles (``model.new_stream()``): for the handles
    whose sessions the C-ABI cannot advance in one call (``ppasr_stream_group_create`` is built for plain Conformer
    handles; Squeezeformer / Efficient-Conformer models keep half-rate layers, grouped attention or a stride layer per
    session).  Same results as driving each session's own stream; N sets of launches per round instead of one."""

    def __init__(self, model, n_sessions, max_frames=0):
        self.model = model
        self.n_sessions = int(n_sessions)
        self._streams = [model.new_stream() for _ in range(self.n_sessions)]

    def offset(self, session):
        return self._streams[int(session)].offset

    def reset(self, session=-1):
        for i in (range(self.n_sessions) if int(session) < 0 else [int(session)]):
            self._streams[i].reset()

    def encode_chunks(self, sessions, speech, want_probs=False):
        m = self.model
        x = torch.as_tensor(speech, dtype=torch.float32).to(m.device)
        assert x.dim() == 3 and int(x.shape[0]) == len(sessions)
        outs = [self._streams[int(sid)].encode_chunk(x[k:k + 1], -16, want_probs=want_probs, want_frames=True)
                for k, sid in enumerate(sessions)]
        fa = torch.cat([o[1] for o in outs], 0)
        fp = torch.cat([o[2] for o in outs], 0)
        if want_probs:
            return fa, fp, torch.cat([o[0] for o in outs], 0)
        return fa, fp


def make_stream_group(model, n_sessions, max_frames=0):
    """``ConformerStreamGroup`` where the library builds session groups for the handle, else ``StreamHandleSet``."""
    try:
        return ConformerStreamGroup(model, n_sessions, max_frames=max_frames)
    except _lib.PPASRHipError as e:
        if e.status != _lib.PPASR_EUNSUPPORTED:
            raise
        return StreamHandleSet(model, n_sessions, max_frames=max_frames)


class ConformerStreamGroup:
    """Many streaming sessions advanced together (no reference counterpart: PPASR streams one session per call,
    predict.py:232-337).  The sessions' K/V and conv caches live in one device allocation;
    ``encode_chunks(sessions, feats)`` advances the listed sessions by one chunk each with ONE set of kernel launches
    (their rows are stacked), so a server's throughput is no longer bound by per-chunk launch overhead.  Every session
    follows ``ConformerStream.encode_chunk(chunk, required_cache_size=-16)`` exactly (full history)."""

    def __init__(self, model, n_sessions, max_frames=0):
        self.model = model
        self.lib = model.lib
        self.n_sessions = int(n_sessions)
        self._g = ctypes.c_void_p()
        with torch.cuda.device(model.device):
            _lib.check(self.lib.ppasr_stream_group_create(model._h, self.n_sessions, int(max_frames),
                                                          ctypes.byref(self._g)))
        self._ws = {}

    def __del__(self):
        g = getattr(self, "_g", None)
        if g is not None and g.value:
            self.lib.ppasr_stream_group_destroy(g)
            self._g = None

    def offset(self, session):
        return int(self.lib.ppasr_stream_group_offset(self._g, int(session)))

    def reset(self, session=-1):
        m = self.model
        with torch.cuda.device(m.device):
            _lib.check(self.lib.ppasr_stream_group_reset(self._g, int(session),
                                                         torch.cuda.current_stream(m.device).cuda_stream))

    def encode_chunks(self, sessions, speech, want_probs=False):
        """sessions: list of distinct slot indices (len n); speech [n,t,F] -> (frame_argmax [n,c] i32, frame_maxprob [n,c])
        device tensors, plus ctc_probs [n,c,V] when ``want_probs``."""
        m = self.model
        x = torch.as_tensor(speech, dtype=torch.float32).to(m.device).contiguous()
        n, T = int(x.shape[0]), int(x.shape[1])
        assert x.dim() == 3 and x.shape[2] == m.input_dim and n == len(sessions)
        c = m.out_frames(T)
        need = int(self.lib.ppasr_group_chunk_workspace_bytes(m._h, n, T))
        key = torch.cuda.current_stream(m.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=m.device)
            self._ws[key] = ws
        probs = torch.empty(n, c, m.vocab_size, dtype=torch.float32, device=m.device) if want_probs else None
        fa = torch.empty(n, c, dtype=torch.int32, device=m.device)
        fp = torch.empty(n, c, dtype=torch.float32, device=m.device)
        ids = (ctypes.c_int * n)(*[int(s) for s in sessions])
        c_out = ctypes.c_int(0)
        with torch.cuda.device(m.device):
            _lib.check(self.lib.ppasr_encode_chunk_group(self._g, ids, n, x.data_ptr(), T,
                                                         None if probs is None else probs.data_ptr(), fa.data_ptr(),
                                                         fp.data_ptr(), ctypes.byref(c_out), ws.data_ptr(), ws.numel(),
                                                         key))
        return (fa, fp, probs) if want_probs else (fa, fp)
