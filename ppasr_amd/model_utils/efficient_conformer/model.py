"""Host-side mirror of ``ppasr/model_utils/efficient_conformer/model.py`` (``EfficientConformerModel``, ctor :17-29,
``get_encoder_out`` :147-161, ``get_encoder_out_chunk`` :163-183), inference surface: grouped attention on
``group_layer_idx`` layers (efficient_conformer/attention.py:128-193), a stride-2 conv layer at ``stride_layer_idx``
(efficient_conformer/encoder.py:455-548; output frame rate 80 ms), 7-tap conv modules after it (encoder.py:123-128).
``get_encoder_out_chunk`` / ``new_stream`` (``EfficientConformerEncoder.forward_chunk``, encoder.py:266-393) are inherited
from the Conformer wrapper: the C-ABI stream object handles this family's cache layout (csrc/capi_stream.hip)."""
import ctypes

import numpy as np
import torch

from ppasr_amd import _lib
from ppasr_amd.model_utils.conformer.model import ConformerModel, _pe_table

__all__ = ["EfficientConformerModel"]


class EfficientConformerModel(ConformerModel):
    def __init__(self, input_dim, vocab_size, mean_istd_path=None, streaming=True, encoder_conf=None,
                 decoder_conf=None, ctc_weight=0.5, state_dict=None, device="cuda:0", **_ignored):
        if state_dict is None:
            raise ValueError("state_dict (Paddle-layout parameter dict) is required")
        if not torch.cuda.is_available():
            raise _lib.PPASRHipError("no HIP device visible: ppasr_amd has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.input_dim, self.vocab_size, self.streaming = input_dim, vocab_size, streaming
        conf = dict(encoder_conf or {})
        eff = dict(conf.get("efficient_conf") or {})
        self.output_size = int(conf.get("output_size", 256))
        self.attention_heads = int(conf.get("attention_heads", 4))
        self.linear_units = int(conf.get("linear_units", 2048))
        self.num_blocks = int(conf.get("num_blocks", 6))
        self.cnn_module_kernel = int(conf.get("cnn_module_kernel", 15))
        self.max_len = int(conf.get("max_len", 5000))

        def one(v, default):
            v = conf.get(v, eff.get(v, default))
            return v

        # stride_layer_idx / stride: an int or a list each (efficient_conformer/encoder.py:50-51,117-121); every stride must be
        # 2 (AvgPool1D(2) residual, kernel // 2 behind it).  One stride layer: the fused kernels and stream handles; several:
        # the library's general layer route, batched encode
        stride_idx = one("stride_layer_idx", 3)
        stride_idx = [] if stride_idx is None else ([int(stride_idx)] if isinstance(stride_idx, int) else [int(v) for v in stride_idx])
        stride = one("stride", 2)
        stride = [int(stride)] * len(stride_idx) if isinstance(stride, int) else [int(v) for v in stride]
        assert len(stride) == len(stride_idx)  # encoder.py:122
        if any(v != 2 for v in stride):
            raise NotImplementedError("stride 2 is built")
        if len(set(stride_idx)) != len(stride_idx) or any(v < 0 or v >= self.num_blocks for v in stride_idx):
            raise ValueError(f"stride_layer_idx={stride_idx}")
        groups = one("group_layer_idx", (0, 1, 2, 3))
        groups = [groups] if isinstance(groups, int) else list(groups or [])
        self.stride_layer_idx = stride_idx[0] if len(stride_idx) == 1 else (None if not stride_idx else list(stride_idx))
        self._stride_layers = list(stride_idx)
        self.group_layer_idx = groups
        self.group_size = int(one("group_size", 3))
        if not one("stride_kernel", True):
            raise NotImplementedError("stride_kernel=False is not built")
        for key, want in (("pos_enc_layer_type", "rel_pos"), ("activation_type", "swish"),
                          ("normalize_before", True), ("use_cnn_module", True)):
            if key in conf and conf[key] != want:
                raise NotImplementedError(f"encoder_conf.{key}={conf[key]!r}: only {want!r} is built")
        # input_layer (encoder.py:93-104): Conv2dSubsampling4, or the 6x / 8x variants (batched encode only)
        il = conf.get("input_layer", "conv2d")
        if il not in ("conv2d", "conv2d6", "conv2d8"):
            raise NotImplementedError(f"encoder_conf.input_layer={il!r}: conv2d, conv2d6 and conv2d8 are built")
        self.input_layer = il
        self.subsampling_rate = {"conv2d": 4, "conv2d6": 6, "conv2d8": 8}[il]
        # cnn_module_norm (convolution.py:65-71): layer_norm, or batch_norm = nn.BatchNorm1D in eval mode, which the
        # library folds into a per-channel scale / shift; the checkpoint carries the running statistics then
        # (default = the reference constructor's: efficient_conformer/encoder.py:49 `cnn_module_norm="batch_norm"`; the
        #  shipped YAML sets layer_norm explicitly, configs/efficient_conformer.yml:9)
        norm = conf.get("cnn_module_norm", "batch_norm")
        if norm not in ("layer_norm", "batch_norm"):
            raise ValueError(f"encoder_conf.cnn_module_norm={norm!r}")
        has_stats = any(k.endswith("conv_module.norm._mean") for k in state_dict)
        if has_stats != (norm == "batch_norm"):
            raise ValueError(f"encoder_conf.cnn_module_norm={norm!r} but the checkpoint "
                             f"{'has' if has_stats else 'lacks'} conv_module.norm._mean / _variance")
        sd = dict(state_dict)
        sd["__pe_table__"] = _pe_table(self.output_size, self.max_len)
        keep = []
        blobs = (_lib.WeightBlob * len(sd))()
        for i, (name, arr) in enumerate(sd.items()):
            a = np.ascontiguousarray(arr, dtype=np.float32)
            keep.append(a)
            blobs[i].name = name.encode()
            blobs[i].data_host = a.ctypes.data
            blobs[i].ndim = min(a.ndim, 4)
            for j in range(min(a.ndim, 4)):
                blobs[i].shape[j] = a.shape[j]
        mask = 0
        for g in groups:
            mask |= 1 << int(g)
        desc = _lib.ModelDesc(_lib.PPASR_MODEL_EFFICIENT_CONFORMER, input_dim, vocab_size, self.output_size,
                              self.attention_heads, self.linear_units, self.num_blocks, self.cnn_module_kernel,
                              1 if streaming else 0,  # causal conv <=> streaming (efficient_conformer/model.py)
                              self.max_len, -1, -1, int(stride_idx[0]) if len(stride_idx) == 1 else -1, mask,
                              self.group_size, 0, {"conv2d": 0, "conv2d6": 6, "conv2d8": 8}[self.input_layer], 0,
                              sum(1 << v for v in stride_idx) if len(stride_idx) > 1 else 0)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ppasr_create(ctypes.byref(desc), blobs, len(sd), ctypes.byref(handle)))
        self._h = handle
        self._ws = None
        self._taps = None
