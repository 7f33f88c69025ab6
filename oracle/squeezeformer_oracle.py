"""ORACLE (test infrastructure, not product code) -- PyTorch-CPU restatement of PPASR's Squeezeformer
encoder + CTC head (ppasr/model_utils/squeezeformer/*).  PARITY UNPINNED (see conformer_oracle.py:
Paddle is not importable offline and the reference ships no tests)."""
import math

import torch
import torch.nn.functional as F

from oracle.conformer_oracle import ConformerOracle


class SqueezeformerOracle(ConformerOracle):
    """SqueezeformerModel.get_encoder_out (squeezeformer/model.py) for the streaming configuration
    (causal conv module, TimeReductionLayerStream; squeezeformer/model.py:35-39)."""

    def __init__(self, sd, attention_heads=4, num_blocks=12, cnn_module_kernel=31, reduce_idx=5, recover_idx=11,
                 max_len=5000, dtype=torch.float32, causal=True, adaptive_scale=True, activation_type="swish",
                 normalize_before=False, pos_enc_layer_type="rel_pos"):
        # causal=False: the non-streaming model (non-causal conv modules, TimeReductionLayer1D; model.py:35-39)
        sd = dict(sd)
        sd.setdefault("encoder.after_norm.weight", sd["encoder.preln.weight"])  # only used for self.d
        super().__init__(sd, attention_heads, num_blocks, cnn_module_kernel, causal, max_len, dtype,
                         activation_type=activation_type)  # (squeezeformer/encoder.py:45,103: FFN and conv module activation)
        self.reduce_idx = reduce_idx
        self.recover_idx = recover_idx
        self.normalize_before = normalize_before  # squeezeformer/encoder.py:49
        self.plain_mha = pos_enc_layer_type != "rel_pos"  # :101-105: conformer's MultiHeadedAttention (no ada scale, no positions)
        # adaptive_scale = False (squeezeformer/encoder.py:44): the parameters exist but are not applied
        # (attention.py:120-123, positionwise.py:63-64, convolution.py:119-120)
        self.adaptive_scale = adaptive_scale
        # dw_stride = True (subsampling.py:38): the second conv of the front end is depthwise (weight [d, 1, 3, 3])
        self.dw_groups = self.d if self.p["encoder.embed.dw_conv.weight"].shape[1] == 1 else 1

    def _ada(self, x, prefix):
        if not self.adaptive_scale:
            return x
        return self.p[prefix + ".ada_scale"].reshape(1, 1, -1) * x + self.p[prefix + ".ada_bias"].reshape(1, 1, -1)

    def ctc_logits(self, enc):
        # final_proj (encoder.py:165-167, 234-235, 381-382): Linear(encoder_dim, output_size) in front of ctc_lo
        if "encoder.final_proj.weight" in self.p:
            enc = enc @ self.p["encoder.final_proj.weight"] + self.p["encoder.final_proj.bias"]
        return enc @ self.p["ctc.ctc_lo.weight"] + self.p["ctc.ctc_lo.bias"]

    def _embed_sq(self, x):
        # DepthwiseConv2DSubsampling4.forward  squeezeformer/subsampling.py:53-68 (dw_stride False -> groups=1)
        x = x.unsqueeze(1)
        x = F.relu(F.conv2d(x, self.p["encoder.embed.pw_conv.weight"], self.p["encoder.embed.pw_conv.bias"], stride=2))
        x = F.relu(F.conv2d(x, self.p["encoder.embed.dw_conv.weight"], self.p["encoder.embed.dw_conv.bias"], stride=2,
                            groups=self.dw_groups))
        b, c, t, f = x.shape
        x = x.permute(0, 2, 1, 3).reshape(b, t, c * f)
        x = x * math.sqrt(self.d)  # RelPositionalEncoding on the c*f-wide tensor (pos_emb not added)
        pos_emb = self.pe[:, 0:t]
        x = self._linear(x, "encoder.embed.input_proj.0")
        return x, pos_emb

    def _attention_sq(self, x, mask, pos_emb, prefix, cache=None):
        # squeezeformer/attention.py:96-162 (adaptive scale :120-123, linear_pos WITH bias :28, cache :128-135)
        if self.plain_mha:  # conformer/attention.py:123-170
            saved, self.pos_type = getattr(self, "pos_type", "rel_pos"), "no_pos"
            try:
                return self._attention(x, mask, pos_emb, cache, prefix)
            finally:
                self.pos_type = saved
        x = self._ada(x, prefix)
        B, T, _ = x.shape
        h, dk = self.h, self.dk
        q = self._linear(x, prefix + ".linear_q").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        k = self._linear(x, prefix + ".linear_k").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        v = self._linear(x, prefix + ".linear_v").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        if cache is not None and cache.shape[0] > 0 and cache.shape[2] > 0:
            key_cache, value_cache = torch.split(cache, dk, dim=-1)
            k = torch.cat([key_cache, k], dim=2)
            v = torch.cat([value_cache, v], dim=2)
        new_cache = torch.cat((k, v), dim=-1)
        p = self._linear(pos_emb, prefix + ".linear_pos").reshape(1, -1, h, dk).permute(0, 2, 1, 3)
        q_u = q + self.p[prefix + ".pos_bias_u"].unsqueeze(1)
        q_v = q + self.p[prefix + ".pos_bias_v"].unsqueeze(1)
        scores = (q_u @ k.transpose(-1, -2) + q_v @ p.transpose(-1, -2)) / math.sqrt(dk)
        if mask is not None and mask.shape[2] > 0:
            m = (mask.unsqueeze(1) == 0)[:, :, :, :scores.shape[-1]]
            scores = scores.masked_fill(m, -float("inf"))
            attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
        else:
            attn = torch.softmax(scores, dim=-1)
        ctx = (attn @ v).permute(0, 2, 1, 3).reshape(B, T, h * dk)
        return self._linear(ctx, prefix + ".linear_out"), new_cache

    def _ffn_sq(self, x, prefix):
        # squeezeformer/positionwise.py:55-65
        x = self._ada(x, prefix)
        return self._linear(self._swish(self._linear(x, prefix + ".w_1")), prefix + ".w_2")

    def _conv_sq(self, x, mask_pad, prefix, cache=None):
        # squeezeformer/convolution.py:102-163 ; mask_pad True = valid here (fill where ~mask_pad)
        x = self._ada(x, prefix)
        x = x.transpose(1, 2)
        x = x.masked_fill(~mask_pad, 0.0)
        if self.lorder > 0:
            if cache is None or cache.shape[-1] == 0:
                x = F.pad(x, (self.lorder, 0), "constant", 0.0)
            else:
                x = torch.cat((cache, x), dim=2)  # the cache holds SCALED inputs (convolution.py:119-137)
            new_cache = x[:, :, -self.lorder:]
        else:
            new_cache = x[:, :, :0]
        x = F.conv1d(x, self.p[prefix + ".pointwise_conv1.weight"], self.p[prefix + ".pointwise_conv1.bias"])
        x = F.glu(x, dim=1)
        if self.trace is not None:
            self.trace[prefix + ".glu"] = x.transpose(1, 2)
        x = F.conv1d(x, self.p[prefix + ".depthwise_conv.weight"], self.p[prefix + ".depthwise_conv.bias"],
                     padding=0 if self.causal else (self.k - 1) // 2, groups=x.shape[1])
        x = x.transpose(1, 2)
        x = self._swish(self._cm_norm(x, prefix + ".norm"))
        x = x.transpose(1, 2)
        x = F.conv1d(x, self.p[prefix + ".pointwise_conv2.weight"], self.p[prefix + ".pointwise_conv2.bias"])
        x = x.masked_fill(~mask_pad, 0.0)
        return x.transpose(1, 2), new_cache

    def _layer_sq(self, i, x, mask, pos_emb, mask_pad, att_cache=None, cnn_cache=None, return_caches=False):
        # SqueezeformerEncoderLayer.forward  squeezeformer/encoder.py:435-506 (normalize_before=False)
        p = f"encoder.encoders.{i}"
        if self.normalize_before:  # :467-493: LayerNorm_k in front of module k, the residual is the un-normalised x
            att, new_att = self._attention_sq(self._ln(x, p + ".layer_norm1"), mask, pos_emb, p + ".self_attn", att_cache)
            x = x + att
            x = x + self._ffn_sq(self._ln(x, p + ".layer_norm2"), p + ".ffn1")
            cv, new_cnn = self._conv_sq(self._ln(x, p + ".layer_norm3"), mask_pad, p + ".conv_module", cnn_cache)
            x = x + cv
            x = x + self._ffn_sq(self._ln(x, p + ".layer_norm4"), p + ".ffn2")
            if return_caches:
                return x, new_att, new_cnn
            return x
        att, new_att = self._attention_sq(x, mask, pos_emb, p + ".self_attn", att_cache)
        x = self._ln(x + att, p + ".layer_norm1")
        x = self._ln(x + self._ffn_sq(x, p + ".ffn1"), p + ".layer_norm2")
        if self.trace is not None:
            self.trace[p + ".x2"] = x
        cv, new_cnn = self._conv_sq(x, mask_pad, p + ".conv_module", cnn_cache)
        x = self._ln(x + cv, p + ".layer_norm3")
        x = self._ln(x + self._ffn_sq(x, p + ".ffn2"), p + ".layer_norm4")
        if return_caches:
            return x, new_att, new_cnn
        return x

    def _time_reduce(self, xs, mask_pad):
        # TimeReductionLayerStream.forward  time_reduction.py:183-206 (kernel 1, no padding) or, for the non-streaming
        # model, TimeReductionLayer1D.forward  :62-85 (kernel 5, padding = kernel - stride = 3 on both sides)
        y = xs.transpose(1, 2).masked_fill(mask_pad == 0, 0.0)
        w = self.p["encoder.time_reduction_layer.dw_conv.weight"]
        y = F.conv1d(y, w, self.p["encoder.time_reduction_layer.dw_conv.bias"], stride=2,
                     padding=max(0, w.shape[-1] - 2), groups=y.shape[1])
        y = F.conv1d(y, self.p["encoder.time_reduction_layer.pw_conv.weight"],
                     self.p["encoder.time_reduction_layer.pw_conv.bias"])
        xs = y.transpose(1, 2)
        mask_pad = mask_pad[:, :, ::2]
        Lr, Tr = mask_pad.shape[-1], xs.shape[1]
        if Lr - Tr < 0:
            xs = xs[:, :Lr - Tr, :]
        elif Lr - Tr > 0:
            xs = torch.cat([xs, torch.zeros(xs.shape[0], Lr - Tr, xs.shape[2], dtype=xs.dtype)], dim=1)
        return xs, mask_pad

    def _factor(self, i):
        # calculate_downsampling_factor  squeezeformer/encoder.py:246-258 (single reduce / recover index)
        red = 1 if (self.reduce_idx is not None and i >= self.reduce_idx) else 0
        rec = 1 if (self.recover_idx is not None and i >= self.recover_idx) else 0
        return int(2 ** (red - rec))

    def forward_chunk(self, xs, offset, required_cache_size, att_cache=None, cnn_cache=None):
        """SqueezeformerEncoder.forward_chunk  squeezeformer/encoder.py:260-381 (B must be 1; att_mask empty)."""
        xs = torch.as_tensor(xs, dtype=self.dtype)
        assert xs.shape[0] == 1
        xs = self._cmvn(xs)
        xs, _ = self._embed_sq(xs)
        cache_t1 = 0 if att_cache is None or att_cache.numel() == 0 else att_cache.shape[2]
        chunk_size = xs.shape[1]
        attention_key_size = cache_t1 + chunk_size
        start = offset - cache_t1
        assert start + attention_key_size < self.max_len
        pos_emb = self.pe[:, start:start + attention_key_size]
        if required_cache_size < 0:
            next_cache_start = 0
        elif required_cache_size == 0:
            next_cache_start = attention_key_size
        else:
            next_cache_start = max(attention_key_size - required_cache_size, 0)
        r_att, r_cnn = [], []
        mask_pad = torch.ones(1, 1, xs.shape[1], dtype=torch.bool)
        max_att_len = 0
        saved = None
        xs = self._ln(xs, "encoder.preln")
        for i in range(self.L):
            if self.reduce_idx is not None and i == self.reduce_idx:
                saved = (xs, pos_emb, mask_pad)
                xs, mask_pad = self._time_reduce(xs, mask_pad)
                pos_emb = pos_emb[:, ::2, :]
            if self.recover_idx is not None and i == self.recover_idx and saved is not None:
                rt, rpos, rpad = saved
                xs = torch.repeat_interleave(xs, 2, dim=1)
                xs = self._linear(xs, "encoder.time_recover_layer")
                xs = rt + xs[:, :rt.shape[1], :]
                pos_emb, mask_pad = rpos, rpad
            factor = self._factor(i)
            if cache_t1 > 0:
                ac = att_cache[i:i + 1][:, :, ::factor, :][:, :, :pos_emb.shape[1] - xs.shape[1], :]
            else:
                ac = None
            cc = None if cnn_cache is None or cnn_cache.numel() == 0 else cnn_cache[i]
            xs, new_att, new_cnn = self._layer_sq(i, xs, None, pos_emb, mask_pad, ac, cc, return_caches=True)
            cached_att = new_att[:, :, next_cache_start // factor:, :]
            cached_att = torch.repeat_interleave(cached_att, factor, dim=2)
            if i == 0:
                max_att_len = cached_att.shape[2]
            r_att.append(cached_att[:, :, :max_att_len, :])
            r_cnn.append(new_cnn.unsqueeze(0))
        return xs, torch.cat(r_att, dim=0), torch.cat(r_cnn, dim=0)

    def get_encoder_out_chunk(self, speech, offset, required_cache_size, att_cache=None, cnn_cache=None):
        """SqueezeformerModel.get_encoder_out_chunk (same shape as conformer/model.py:164-184)."""
        with torch.no_grad():
            xs, att_cache, cnn_cache = self.forward_chunk(speech, offset, required_cache_size, att_cache, cnn_cache)
            probs = torch.softmax(self.ctc_logits(xs), dim=2)
        return probs, att_cache, cnn_cache

    def encoder_forward(self, speech, speech_lengths, return_layers=False):
        # SqueezeformerEncoder.forward  squeezeformer/encoder.py:172-236
        xs = torch.as_tensor(speech, dtype=self.dtype)
        lens = torch.as_tensor(speech_lengths, dtype=torch.int64)
        T = xs.shape[1]
        masks = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(1)
        xs = self._cmvn(xs)
        xs, pos_emb = self._embed_sq(xs)
        masks = masks[:, :, :-2:2][:, :, :-2:2]
        mask_pad = masks
        chunk_masks = masks & torch.ones(1, xs.shape[1], xs.shape[1], dtype=torch.bool)
        xs = self._ln(xs, "encoder.preln")
        layers = [xs]
        saved = None
        for i in range(self.L):
            if self.reduce_idx is not None and i == self.reduce_idx:
                saved = (xs, chunk_masks, pos_emb, mask_pad)
                xs, mask_pad = self._time_reduce(xs, mask_pad)
                chunk_masks = chunk_masks[:, ::2, ::2]
                pos_emb = pos_emb[:, ::2, :]
            if self.recover_idx is not None and i == self.recover_idx and saved is not None:
                rt, rmask, rpos, rpad = saved
                xs = torch.repeat_interleave(xs, 2, dim=1)
                xs = self._linear(xs, "encoder.time_recover_layer")
                xs = rt + xs[:, :rt.shape[1], :]
                chunk_masks, pos_emb, mask_pad = rmask, rpos, rpad
            xs = self._layer_sq(i, xs, chunk_masks, pos_emb, mask_pad)
            layers.append(xs)
        if return_layers:
            return xs, masks, layers
        return xs, masks
