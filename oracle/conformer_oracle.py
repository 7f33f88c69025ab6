"""ORACLE (test infrastructure, not product code) -- PyTorch-CPU fp32/fp64
restatement of PPASR's Conformer encoder + CTC head.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``ppasr_amd``) never does.

PARITY UNPINNED: PaddlePaddle is not installable in the build container and the
reference ships no tests / golden vectors, so this file is a
line-by-line restatement of the reference sources, pinned only by its own
committed goldens (``tests/golden/``).  Every function cites the reference
``file:line`` it follows (paths relative to ``/root/reference/ppasr``).

Parameters come in as a dict of numpy arrays with the Paddle names / layouts
(``Linear.weight`` is ``[in, out]`` -> ``x @ W + b``).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, name, dtype):
    return torch.from_numpy(np.ascontiguousarray(sd[name])).to(dtype)


_ACTIVATIONS = {
    "swish": lambda x: x * torch.sigmoid(x),  # utils/common.py:201 (paddle.nn.Swish)
    "relu": torch.relu,
    "relu6": lambda x: torch.clamp(x, 0.0, 6.0),
    "tanh": torch.tanh,
    "gelu": lambda x: F.gelu(x),  # paddle.nn.GELU(approximate=False)
    "elu": lambda x: F.elu(x, 1.0),
    "selu": torch.selu,
    "leakyrelu": lambda x: F.leaky_relu(x, 0.01),
    "hardtanh": lambda x: torch.clamp(x, -1.0, 1.0),
    "hardswish": F.hardswish,
    "hardshrink": lambda x: F.hardshrink(x, 0.5),
}


class ConformerOracle:
    """Functional restatement of ``ConformerModel`` (model_utils/conformer/model.py:16)
    for inference: ``get_encoder_out`` (:148) and ``get_encoder_out_chunk`` (:164)."""

    def __init__(self, sd, attention_heads=4, num_blocks=12, cnn_module_kernel=15, causal=True,
                 max_len=5000, dtype=torch.float32, pos_enc_layer_type="rel_pos", normalize_before=True,
                 concat_after=False, macaron_style=True, use_cnn_module=True, activation_type="swish"):
        # the remaining ConformerEncoder.__init__ arguments (conformer/encoder.py:38-48); input_layer is read off the
        # parameter names (_sub_kind)
        self.pos_type = pos_enc_layer_type
        self.pre_norm = normalize_before
        self.concat_after = concat_after
        self.macaron = macaron_style
        self.use_cnn = use_cnn_module
        self.act = _ACTIVATIONS[activation_type]  # get_activation (utils/common.py:189-206), Paddle's default parameters
        self.ff_scale = 0.5 if macaron_style else 1.0  # encoder.py:330-334
        self.dtype = dtype
        self.h = attention_heads
        self.L = num_blocks
        self.k = cnn_module_kernel
        # causal conv <=> streaming model (conformer/model.py:35-39)
        self.lorder = cnn_module_kernel - 1 if causal else 0
        self.causal = causal
        self.p = {k: _t(sd, k, dtype) for k in sd}
        self.d = self.p["encoder.after_norm.weight"].shape[0]
        self.dk = self.d // self.h
        self.max_len = max_len
        self.trace = None  # set to a dict to record per-layer intermediates (kernel-level parity tests)
        # PositionalEncoding.__init__  conformer/embedding.py:38-53 (table built in fp32)
        pe = torch.zeros(max_len, self.d, dtype=torch.float32)
        position = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, self.d, 2, dtype=torch.float32) * -(math.log(10000.0) / self.d))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.pe = pe.to(dtype).unsqueeze(0)  # [1, max_len, d]

    # ---- shared primitives ------------------------------------------------
    def _linear(self, x, prefix, bias=True):
        # paddle Linear: x @ W[in,out] + b   (utils/base.py:58)
        y = x @ self.p[prefix + ".weight"]
        if bias:
            y = y + self.p[prefix + ".bias"]
        return y

    def _ln(self, x, prefix, eps=1e-5):
        # nn.LayerNorm, biased variance, eps=1e-5 (utils/base.py:7; encoder.py:327-336)
        return F.layer_norm(x, (x.shape[-1],), self.p[prefix + ".weight"], self.p[prefix + ".bias"], eps)

    def _cm_norm(self, x, prefix, eps=1e-5):
        """ConvolutionModule.norm on [B, T, C]: nn.LayerNorm(channels), or -- cnn_module_norm: batch_norm
        (convolution.py:65-71) -- nn.BatchNorm1D(channels) in eval mode: the per-channel running statistics
        (parameters `_mean`, `_variance`; epsilon 1e-5), the same for every frame."""
        if prefix + "._mean" in self.p:
            inv = torch.rsqrt(self.p[prefix + "._variance"] + eps)
            return (x - self.p[prefix + "._mean"]) * inv * self.p[prefix + ".weight"] + self.p[prefix + ".bias"]
        return self._ln(x, prefix, eps)

    def _swish(self, x):
        return self.act(x)  # `activation` of PositionwiseFeedForward / ConvolutionModule (swish in every shipped YAML)

    # ---- embed --------------------------------------------------------------
    def _cmvn(self, x):
        # GlobalCMVN.forward  utils/cmvn.py:29-31
        return (x - self.p["encoder.global_cmvn.mean"]) * self.p["encoder.global_cmvn.istd"]

    def _sub_kind(self):
        """4, 6 or 8: which Conv2dSubsampling class the parameters belong to (subsampling.py:63, 118, 160): the 6x / 8x
        classes name their projection `linear`, the 8x one has a third conv (`conv.4`)."""
        if "encoder.embed.out.1.weight" in self.p:
            return 1  # LinearNoSubsampling (subsampling.py:24-65): out = Sequential(Linear, LayerNorm, Dropout, ReLU)
        if "encoder.embed.out.0.weight" in self.p:
            return 4
        return 8 if "encoder.embed.conv.4.weight" in self.p else 6

    def _sub_masks(self, masks):
        """The mask slicing of the subsampling classes (subsampling.py:115, 157, 205)."""
        k = self._sub_kind()
        if k == 1:
            return masks
        if k == 4:
            return masks[:, :, :-2:2][:, :, :-2:2]
        if k == 6:
            return masks[:, :, :-2:2][:, :, :-4:3]
        return masks[:, :, :-2:2][:, :, :-2:2][:, :, :-2:2]

    def _embed(self, x, offset):
        # Conv2dSubsampling4 / 6 / 8 .forward  conformer/subsampling.py:96-115, 144-157, 191-205
        k = self._sub_kind()
        if k == 1:
            x = F.relu(self._ln(self._linear(x, "encoder.embed.out.0"), "encoder.embed.out.1", eps=1e-12))
            t = x.shape[1]
        else:
            x = x.unsqueeze(1)
            x = F.relu(F.conv2d(x, self.p["encoder.embed.conv.0.weight"], self.p["encoder.embed.conv.0.bias"], stride=2))
            x = F.relu(F.conv2d(x, self.p["encoder.embed.conv.2.weight"], self.p["encoder.embed.conv.2.bias"],
                                stride=3 if k == 6 else 2))
            if k == 8:
                x = F.relu(F.conv2d(x, self.p["encoder.embed.conv.4.weight"], self.p["encoder.embed.conv.4.bias"], stride=2))
            b, c, t, f = x.shape
            x = self._linear(x.permute(0, 2, 1, 3).reshape(b, t, c * f),
                             "encoder.embed.out.0" if k == 4 else "encoder.embed.linear")
        if self.pos_type == "no_pos":  # NoPositionalEncoding.forward  embedding.py:18-19
            return x, None
        assert offset + t < self.max_len
        pos_emb = self.pe[:, offset:offset + t]
        if self.pos_type == "abs_pos":  # PositionalEncoding.forward  embedding.py:55-72
            return x * math.sqrt(self.d) + pos_emb, pos_emb
        # RelPositionalEncoding.forward  conformer/embedding.py:102-115 (x*sqrt(d); pos_emb NOT added)
        return x * math.sqrt(self.d), pos_emb

    def conv_intermediates(self, speech):
        """conv1 / conv2 activations (NCHW) for kernel-level parity tests."""
        x = self._cmvn(torch.as_tensor(speech, dtype=self.dtype)).unsqueeze(1)
        y1 = F.relu(F.conv2d(x, self.p["encoder.embed.conv.0.weight"], self.p["encoder.embed.conv.0.bias"], stride=2))
        y2 = F.relu(F.conv2d(y1, self.p["encoder.embed.conv.2.weight"], self.p["encoder.embed.conv.2.bias"], stride=2))
        return y1, y2

    # ---- one encoder layer ----------------------------------------------------
    def _ffn(self, x, prefix):
        # PositionwiseFeedForward.forward  conformer/positionwise.py:32-39
        return self._linear(self._swish(self._linear(x, prefix + ".w_1")), prefix + ".w_2")

    def _attention(self, x, mask, pos_emb, cache, prefix):
        # RelPositionMultiHeadedAttention.forward  conformer/attention.py:198-262
        B, T, _ = x.shape
        h, dk = self.h, self.dk
        q = self._linear(x, prefix + ".linear_q").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        k = self._linear(x, prefix + ".linear_k").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        v = self._linear(x, prefix + ".linear_v").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        if cache is not None and cache.shape[0] > 0:  # :225-229
            key_cache, value_cache = torch.split(cache, dk, dim=-1)
            k = torch.cat([key_cache, k], dim=2)
            v = torch.cat([value_cache, v], dim=2)
        new_cache = torch.cat((k, v), dim=-1)  # :232
        if self.pos_type == "rel_pos":
            p = self._linear(pos_emb, prefix + ".linear_pos", bias=False)
            p = p.reshape(pos_emb.shape[0], -1, h, dk).permute(0, 2, 1, 3)  # :234-236
            q_u = q + self.p[prefix + ".pos_bias_u"].unsqueeze(1)  # :241
            q_v = q + self.p[prefix + ".pos_bias_v"].unsqueeze(1)  # :243
            matrix_ac = q_u @ k.transpose(-1, -2)  # :250
            matrix_bd = q_v @ p.transpose(-1, -2)  # :255  (rel_shift disabled :256-258)
            scores = (matrix_ac + matrix_bd) / math.sqrt(dk)  # :260
        else:  # MultiHeadedAttention.forward  conformer/attention.py:123-170 (abs_pos / no_pos)
            scores = (q @ k.transpose(-1, -2)) / math.sqrt(dk)
        # MultiHeadedAttention.forward_attention  conformer/attention.py:86-126
        if mask is not None and mask.shape[2] > 0:
            m = (mask.unsqueeze(1) == 0)[:, :, :, :scores.shape[-1]]
            scores = scores.masked_fill(m, -float("inf"))
            attn = torch.softmax(scores, dim=-1)
            attn = attn.masked_fill(m, 0.0)  # also overwrites NaN rows (fully masked)
        else:
            attn = torch.softmax(scores, dim=-1)
        ctx = (attn @ v).permute(0, 2, 1, 3).reshape(B, T, h * dk)
        if self.trace is not None:
            self.trace[prefix + ".q"] = q.permute(0, 2, 1, 3).reshape(B, T, h * dk)
            self.trace[prefix + ".k"] = k.permute(0, 2, 1, 3).reshape(B, -1, h * dk)
            self.trace[prefix + ".v"] = v.permute(0, 2, 1, 3).reshape(B, -1, h * dk)
            self.trace[prefix + ".ctx"] = ctx
        return self._linear(ctx, prefix + ".linear_out"), new_cache

    def _conv_module(self, x, mask_pad, cache, prefix):
        # ConvolutionModule.forward  conformer/convolution.py:82-143
        # mask_pad: bool [B,1,T], True = PAD (encoder.py:192 passes ~masks)
        x = x.transpose(1, 2)  # [B, C, T]
        if mask_pad is not None and mask_pad.shape[2] > 0:
            x = x.masked_fill(mask_pad, 0.0)
        if self.lorder > 0:
            if cache is None or cache.shape[2] == 0:
                x = F.pad(x, (self.lorder, 0), "constant", 0.0)
            else:
                x = torch.cat((cache, x), dim=2)
            new_cache = x[:, :, -self.lorder:]
        else:
            new_cache = torch.zeros(0, 0, 0, dtype=x.dtype)
        x = F.conv1d(x, self.p[prefix + ".pointwise_conv1.weight"], self.p[prefix + ".pointwise_conv1.bias"])
        x = F.glu(x, dim=1)
        if self.trace is not None:
            self.trace[prefix + ".glu"] = x.transpose(1, 2)  # [B, lorder+T, C]
        pad = 0 if self.lorder > 0 else (self.k - 1) // 2
        x = F.conv1d(x, self.p[prefix + ".depthwise_conv.weight"], self.p[prefix + ".depthwise_conv.bias"],
                     padding=pad, groups=x.shape[1])
        x = x.transpose(1, 2)
        x = self._swish(self._cm_norm(x, prefix + ".norm"))  # LayerNorm or BatchNorm1D(eval) (convolution.py:65-71)
        x = x.transpose(1, 2)
        x = F.conv1d(x, self.p[prefix + ".pointwise_conv2.weight"], self.p[prefix + ".pointwise_conv2.bias"])
        if mask_pad is not None and mask_pad.shape[2] > 0:
            x = x.masked_fill(mask_pad, 0.0)
        return x.transpose(1, 2), new_cache

    def _layer(self, i, x, mask, pos_emb, mask_pad, att_cache=None, cnn_cache=None):
        # ConformerEncoderLayer.forward  conformer/encoder.py:346-431
        p = f"encoder.encoders.{i}"
        pre = self.pre_norm
        if self.macaron:
            residual = x
            if pre:
                x = self._ln(x, p + ".norm_ff_macaron")
            x = residual + self.ff_scale * self._ffn(x, p + ".feed_forward_macaron")
            if not pre:
                x = self._ln(x, p + ".norm_ff_macaron")
        if self.trace is not None:
            self.trace[p + ".x1"] = x
        residual = x
        if pre:
            x = self._ln(x, p + ".norm_mha")
        x_att, new_att_cache = self._attention(x, mask, pos_emb, att_cache, p + ".self_attn")
        if self.concat_after:  # :395-397
            x = residual + self._linear(torch.cat((x, x_att), dim=-1), p + ".concat_linear")
        else:
            x = residual + x_att
        if not pre:
            x = self._ln(x, p + ".norm_mha")
        if self.trace is not None:
            self.trace[p + ".x2"] = x
        new_cnn_cache = torch.zeros(0, 0, 0, dtype=x.dtype)  # :405
        if self.use_cnn:
            residual = x
            if pre:
                x = self._ln(x, p + ".norm_conv")
            x, new_cnn_cache = self._conv_module(x, mask_pad, cnn_cache, p + ".conv_module")
            x = residual + x
            if not pre:
                x = self._ln(x, p + ".norm_conv")
        residual = x
        if pre:
            x = self._ln(x, p + ".norm_ff")
        x = residual + self.ff_scale * self._ffn(x, p + ".feed_forward")
        if not pre:
            x = self._ln(x, p + ".norm_ff")
        if self.use_cnn:
            x = self._ln(x, p + ".norm_final")
        return x, new_att_cache, new_cnn_cache

    # ---- encoder ------------------------------------------------------------------
    def encoder_forward(self, speech, speech_lengths, return_layers=False):
        """ConformerEncoder.forward with decoding_chunk_size=-1  (conformer/encoder.py:164-206)."""
        xs = torch.as_tensor(speech, dtype=self.dtype)
        lens = torch.as_tensor(speech_lengths, dtype=torch.int64)
        T = xs.shape[1]
        # make_non_pad_mask  utils/mask.py:22-67  (max_len = lengths.max(); the padded batch has T == max)
        masks = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(1)  # [B,1,T] True=valid
        xs = self._cmvn(xs)
        xs, pos_emb = self._embed(xs, 0)
        masks = self._sub_masks(masks)  # subsampling.py:115 / 157 / 205
        mask_pad = ~masks
        # add_optional_chunk_mask with decoding_chunk_size<0: chunk = max_len -> all-ones & pad mask
        # (utils/mask.py:153-177) -> [B, T', T'] ; only keys are masked
        chunk_masks = masks & torch.ones(1, xs.shape[1], xs.shape[1], dtype=torch.bool)
        layers = [xs]
        for i in range(self.L):
            xs, _, _ = self._layer(i, xs, chunk_masks, pos_emb, mask_pad)
            layers.append(xs)
        if self.pre_norm:  # :201
            xs = self._ln(xs, "encoder.after_norm")
        if return_layers:
            return xs, masks, layers
        return xs, masks

    def ctc_logits(self, enc):
        return enc @ self.p["ctc.ctc_lo.weight"] + self.p["ctc.ctc_lo.bias"]

    def get_encoder_out(self, speech, speech_lengths, return_logits=False):
        """ConformerModel.get_encoder_out  conformer/model.py:148-162 -> softmax(ctc_lo(enc)) (loss/ctc.py:62-70)."""
        with torch.no_grad():
            enc, _ = self.encoder_forward(speech, speech_lengths)
            logits = self.ctc_logits(enc)
            probs = torch.softmax(logits, dim=2)
        if return_logits:
            return probs, logits
        return probs

    def forward_chunk(self, xs, offset, required_cache_size, att_cache=None, cnn_cache=None):
        """ConformerEncoder.forward_chunk  conformer/encoder.py:208-283 (B must be 1)."""
        xs = torch.as_tensor(xs, dtype=self.dtype)
        assert xs.shape[0] == 1
        xs = self._cmvn(xs)
        xs, _ = self._embed(xs, offset)
        cache_t1 = 0 if att_cache is None or att_cache.numel() == 0 else att_cache.shape[2]
        chunk_size = xs.shape[1]
        attention_key_size = cache_t1 + chunk_size
        start = offset - cache_t1
        pos_emb = None
        if self.pos_type != "no_pos":
            assert start + attention_key_size < self.max_len  # embedding.py:84
            pos_emb = self.pe[:, start:start + attention_key_size]  # :253
        if required_cache_size < 0:
            next_cache_start = 0
        elif required_cache_size == 0:
            next_cache_start = attention_key_size
        else:
            next_cache_start = max(attention_key_size - required_cache_size, 0)
        r_att, r_cnn = [], []
        for i in range(self.L):
            ac = None if cache_t1 == 0 else att_cache[i:i + 1]
            cc = None if cnn_cache is None or cnn_cache.numel() == 0 else cnn_cache[i]
            xs, new_att, new_cnn = self._layer(i, xs, None, pos_emb, None, ac, cc)
            r_att.append(new_att[:, :, next_cache_start:, :])
            r_cnn.append(new_cnn)
        if self.pre_norm:  # :276
            xs = self._ln(xs, "encoder.after_norm")
        return xs, torch.cat(r_att, dim=0), torch.stack(r_cnn, dim=0)

    def get_encoder_out_chunk(self, speech, offset, required_cache_size, att_cache=None, cnn_cache=None):
        """ConformerModel.get_encoder_out_chunk  conformer/model.py:164-184."""
        with torch.no_grad():
            xs, att_cache, cnn_cache = self.forward_chunk(speech, offset, required_cache_size, att_cache, cnn_cache)
            probs = torch.softmax(self.ctc_logits(xs), dim=2)
        return probs, att_cache, cnn_cache
