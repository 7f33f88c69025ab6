"""Host-side mirror of ``ppasr/model_utils/squeezeformer/model.py`` (``SqueezeformerModel``, ctor :17-29,
``get_encoder_out`` :156-170, ``get_encoder_out_chunk`` :172-192), inference surface: post-LN blocks with adaptive scale
(squeezeformer/encoder.py:435-506), time reduction / recovery around ``reduce_idx`` / ``recover_idx``
(encoder.py:210-230, time_reduction.py:183-206), depthwise-conv subsampling (subsampling.py:53-68).
``get_encoder_out_chunk`` / ``new_stream`` (``SqueezeformerEncoder.forward_chunk``, encoder.py:260-381) are inherited from
the Conformer wrapper: the C-ABI stream object handles this family's cache layout (csrc/capi_stream.hip).  Shares the
C-ABI plumbing with the Conformer wrapper; only the model descriptor and parameter names differ."""
import ctypes

import numpy as np
import torch

from ppasr_amd import _lib
from ppasr_amd.model_utils.conformer.model import ConformerModel, _pe_table

__all__ = ["SqueezeformerModel"]


class SqueezeformerModel(ConformerModel):
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
        self.output_size = int(conf.get("encoder_dim", 256))
        # output_size != encoder_dim: the reference appends final_proj = Linear(encoder_dim, output_size) (encoder.py:165-167);
        # ppasr_create folds it into the CTC head when the checkpoint carries encoder.final_proj.weight
        self.final_output_size = int(conf.get("output_size", self.output_size))
        if (self.final_output_size != self.output_size) != ("encoder.final_proj.weight" in state_dict):
            raise ValueError("encoder_conf.output_size != encoder_dim <=> the checkpoint has encoder.final_proj.*")
        self.attention_heads = int(conf.get("attention_heads", 4))
        self.linear_units = self.output_size * int(conf.get("feed_forward_expansion_factor", 8))
        self.num_blocks = int(conf.get("num_blocks", 12))
        self.cnn_module_kernel = int(conf.get("cnn_module_kernel", 31))
        self.max_len = int(conf.get("max_len", 5000))
        self.reduce_idx = conf.get("reduce_idx", 5)
        self.recover_idx = conf.get("recover_idx", 11)
        # pos_enc_layer_type (squeezeformer/encoder.py:101-112): "rel_pos", or anything else = conformer's plain
        # MultiHeadedAttention (the front end keeps its RelPositionalEncoding scaling either way); general route
        self.pos_enc_layer_type = str(conf.get("pos_enc_layer_type", "rel_pos"))
        # normalize_before = True (squeezeformer/encoder.py:49,467-493): LayerNorm in front of every module; general route
        self.normalize_before = bool(conf.get("normalize_before", False))
        # activation_type (squeezeformer/encoder.py:45: the feed-forward modules' and the conv module's activation): anything
        # but swish runs on the library's general layer route
        from ppasr_amd.model_utils.conformer.model import _ACT_CODES
        act = conf.get("activation_type", "swish")
        if act not in _ACT_CODES:
            raise KeyError(act)  # get_activation (utils/common.py:206)
        self.activation_type = act
        # adaptive_scale = False: a flag (the checkpoint still carries the unused ada_scale / ada_bias, attention.py:34-37);
        # dw_stride = True: recognised by ppasr_create from the depthwise shape of encoder.embed.dw_conv.weight
        self.adaptive_scale = bool(conf.get("adaptive_scale", True))
        dw = bool(conf.get("dw_stride", False))
        if "encoder.embed.dw_conv.weight" in state_dict and (np.asarray(state_dict["encoder.embed.dw_conv.weight"]).shape[1] == 1) != dw:
            raise ValueError(f"encoder_conf.dw_stride={dw} does not match the shape of encoder.embed.dw_conv.weight")
        # cnn_norm_type (squeezeformer/encoder.py:41): layer_norm, or batch_norm = BatchNorm1D in eval mode (folded)
        norm = conf.get("cnn_norm_type", "layer_norm")
        if norm not in ("layer_norm", "batch_norm"):
            raise ValueError(f"encoder_conf.cnn_norm_type={norm!r}")
        has_stats = any(k.endswith("conv_module.norm._mean") for k in state_dict)
        if has_stats != (norm == "batch_norm"):
            raise ValueError(f"encoder_conf.cnn_norm_type={norm!r} but the checkpoint "
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
        desc = _lib.ModelDesc(_lib.PPASR_MODEL_SQUEEZEFORMER, input_dim, vocab_size, self.output_size,
                              self.attention_heads, self.linear_units, self.num_blocks, self.cnn_module_kernel,
                              1 if streaming else 0,  # causal conv + stream time-reduction <=> streaming (model.py:35-39)
                              self.max_len, -1 if self.reduce_idx is None else int(self.reduce_idx),
                              -1 if self.recover_idx is None else int(self.recover_idx), -1, 0, 0)
        if not self.adaptive_scale:
            desc.options |= _lib.PPASR_OPT_SQ_NO_ADAPTIVE_SCALE
        desc.options |= _ACT_CODES[act] << _lib.PPASR_OPT_ACT_SHIFT
        if self.normalize_before:
            desc.options |= _lib.PPASR_OPT_SQ_PRE_NORM
        if self.pos_enc_layer_type != "rel_pos":
            desc.options |= 2  # PPASR_OPT_POS_NONE: no positional term in the attention scores
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ppasr_create(ctypes.byref(desc), blobs, len(sd), ctypes.byref(handle)))
        self._h = handle
        self._ws = None
        self._taps = None
