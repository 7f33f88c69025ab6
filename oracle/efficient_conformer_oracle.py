"""ORACLE (test infrastructure, not product code) -- PyTorch-CPU restatement of PPASR's
Efficient-Conformer encoder + CTC head (ppasr/model_utils/efficient_conformer/*).  PARITY UNPINNED
(Paddle is not importable offline; the reference ships no tests)."""
import math

import torch
import torch.nn.functional as F

from oracle.conformer_oracle import ConformerOracle


class EfficientConformerOracle(ConformerOracle):
    """EfficientConformerModel.get_encoder_out for the streaming configuration (causal conv)."""

    def __init__(self, sd, attention_heads=4, num_blocks=12, cnn_module_kernel=15, stride_layer_idx=3,
                 group_layer_idx=(0, 1, 2, 3), group_size=3, max_len=5000, dtype=torch.float32, causal=True):
        # causal conv <=> streaming model (efficient_conformer/model.py: causal = streaming)
        super().__init__(sd, attention_heads, num_blocks, cnn_module_kernel, causal, max_len, dtype)
        self.stride_layer_idx = stride_layer_idx
        self.group_layer_idx = tuple(group_layer_idx or ())
        self.group_size = group_size

    def _kernel(self, i):
        # cnn_module_kernels: halves after each stride layer (encoder.py:123-128, stride_kernel=True)
        return self.k >> sum(1 for v in self._strides() if v < i)

    def _strides(self):
        s = self.stride_layer_idx
        return [] if s is None else ([s] if isinstance(s, int) else list(s))

    def _grouped_attention(self, x, mask, pos_emb, prefix, cache=None):
        # GroupedRelPositionMultiHeadedAttention.forward  efficient_conformer/attention.py:128-193
        B, T, _ = x.shape
        h, dk, g = self.h, self.dk, self.group_size
        q = self._linear(x, prefix + ".linear_q").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        k = self._linear(x, prefix + ".linear_k").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        v = self._linear(x, prefix + ".linear_v").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        p = self._linear(pos_emb, prefix + ".linear_pos")  # with bias (:31)
        if cache is not None and cache.shape[0] > 0 and cache.shape[2] > 0:  # :155-158
            key_cache, value_cache = torch.split(cache, dk, dim=-1)
            k = torch.cat([key_cache, k], dim=2)
            v = torch.cat([value_cache, v], dim=2)
        new_cache = torch.cat((k, v), dim=-1)  # before grouping (:161)
        if mask is not None and mask.shape[2] > 0:  # :164-167
            time2 = mask.shape[2]
            k = k[:, :, -time2:, :]
            v = v[:, :, -time2:, :]
        # pad4group :40-79 (queries and keys are padded separately)
        pad_t = (g - T % g) % g
        pad_kv = (g - k.shape[2] % g) % g
        q = F.pad(q, (0, 0, 0, pad_t))
        k = F.pad(k, (0, 0, 0, pad_kv))
        v = F.pad(v, (0, 0, 0, pad_kv))
        if mask is not None and mask.shape[2] > 0:
            mask = mask[:, ::g, ::g]
        q = q.permute(0, 2, 1, 3).reshape(B, -1, h, dk * g).permute(0, 2, 1, 3)
        k = k.permute(0, 2, 1, 3).reshape(B, -1, h, dk * g).permute(0, 2, 1, 3)
        v = v.permute(0, 2, 1, 3).reshape(B, -1, h, dk * g).permute(0, 2, 1, 3)
        pad_p = (g - p.shape[1] % g) % g
        p = F.pad(p, (0, 0, 0, pad_p)).reshape(p.shape[0], -1, h, dk * g).permute(0, 2, 1, 3)
        q = q.permute(0, 2, 1, 3)
        q_u = (q + self.p[prefix + ".pos_bias_u"]).permute(0, 2, 1, 3)
        q_v = (q + self.p[prefix + ".pos_bias_v"]).permute(0, 2, 1, 3)
        scores = (q_u @ k.transpose(-1, -2) + q_v @ p.transpose(-1, -2)) / math.sqrt(dk * g)
        if mask is not None and mask.shape[2] > 0:
            m = (mask.unsqueeze(1) == 0)[:, :, :, :scores.shape[-1]]
            scores = scores.masked_fill(m, -float("inf"))
            attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
        else:
            attn = torch.softmax(scores, dim=-1)
        ctx = (attn @ v).permute(0, 2, 1, 3).reshape(B, -1, h * dk)
        ctx = ctx[:, :ctx.shape[1] - pad_t]
        if self.trace is not None:
            self.trace[prefix + ".ctx"] = ctx
        return self._linear(ctx, prefix + ".linear_out"), new_cache

    def _conv_eff(self, x, mask_pad, prefix, ksize, stride, cache=None):
        # efficient_conformer/convolution.py:80-138 ; mask_pad True = valid
        lorder = ksize - 1 if self.causal else 0  # convolution.py ctor: causal -> lorder = k - 1, padding 0;
        padding = 0 if self.causal else (ksize - 1) // 2  # else lorder 0, depthwise padding (k - 1) // 2
        x = x.transpose(1, 2).masked_fill(~mask_pad, 0.0)
        if lorder > 0:
            if cache is None or cache.shape[-1] == 0:
                x = F.pad(x, (lorder, 0), "constant", 0.0)
            else:
                x = torch.cat((cache[:, :, -lorder:], x), dim=2)  # :105-109
            new_cache = x[:, :, -lorder:]
        else:
            new_cache = x[:, :, :0]
        x = F.conv1d(x, self.p[prefix + ".pointwise_conv1.weight"], self.p[prefix + ".pointwise_conv1.bias"])
        x = F.glu(x, dim=1)
        x = F.conv1d(x, self.p[prefix + ".depthwise_conv.weight"], self.p[prefix + ".depthwise_conv.bias"],
                     stride=stride, padding=padding, groups=x.shape[1])
        x = x.transpose(1, 2)
        x = self._swish(self._cm_norm(x, prefix + ".norm"))
        x = x.transpose(1, 2)
        x = F.conv1d(x, self.p[prefix + ".pointwise_conv2.weight"], self.p[prefix + ".pointwise_conv2.bias"])
        if mask_pad.shape[2] != x.shape[2]:
            mask_pad = mask_pad[:, :, ::stride]
        x = x.masked_fill(~mask_pad, 0.0)
        return x.transpose(1, 2), new_cache

    def _layer_eff(self, i, x, mask, pos_emb, mask_pad, att_cache=None, cnn_cache=None, return_caches=False):
        # ConformerEncoderLayer / StrideConformerEncoderLayer (efficient_conformer/encoder.py:455-548)
        p = f"encoder.encoders.{i}"
        x = x + 0.5 * self._ffn(self._ln(x, p + ".norm_ff_macaron"), p + ".feed_forward_macaron")
        xn = self._ln(x, p + ".norm_mha")
        if i in self.group_layer_idx:
            x_att, new_att = self._grouped_attention(xn, mask, pos_emb, p + ".self_attn", att_cache)
        else:
            x_att, new_att = self._attention(xn, mask, pos_emb, att_cache, p + ".self_attn")
        x = x + x_att
        residual = x
        stride = 2 if i in self._strides() else 1
        y, new_cnn = self._conv_eff(self._ln(x, p + ".norm_conv"), mask_pad, p + ".conv_module", self._kernel(i), stride,
                                    cnn_cache)
        if stride > 1:
            # paddle.nn.AvgPool1D(2, 2, padding=0, ceil_mode=True), exclusive (encoder.py:171-172)
            residual = F.avg_pool1d(residual.transpose(1, 2), kernel_size=2, stride=2, padding=0, ceil_mode=True,
                                    count_include_pad=False).transpose(1, 2)
        x = residual + y
        x = x + 0.5 * self._ffn(self._ln(x, p + ".norm_ff"), p + ".feed_forward")
        x = self._ln(x, p + ".norm_final")
        if return_caches:
            return x, new_att, new_cnn
        return x

    def _factor(self, i):
        # calculate_downsampling_factor  efficient_conformer/encoder.py:205-210
        return 2 ** sum(1 for v in self._strides() if v < i)

    def forward_chunk(self, xs, offset, required_cache_size, att_cache=None, cnn_cache=None):
        """EfficientConformerEncoder.forward_chunk  efficient_conformer/encoder.py:266-393 (B = 1, empty att_mask,
        global_chunk_size = 0)."""
        xs = torch.as_tensor(xs, dtype=self.dtype)
        assert xs.shape[0] == 1
        offset = offset * self._factor(self.L + 1)  # :305
        xs = self._cmvn(xs)
        xs, _ = self._embed(xs, offset)
        cache_t1 = 0 if att_cache is None or att_cache.numel() == 0 else att_cache.shape[2]
        chunk_size = xs.shape[1]
        attention_key_size = cache_t1 + chunk_size
        start = offset - cache_t1
        assert start >= 0 and start + attention_key_size < self.max_len
        pos_emb = self.pe[:, start:start + attention_key_size]
        if required_cache_size < 0:
            next_cache_start = 0
        elif required_cache_size == 0:
            next_cache_start = attention_key_size
        else:
            next_cache_start = max(attention_key_size - required_cache_size, 0)
        r_att, r_cnn = [], []
        mask_pad = torch.ones(1, 1, xs.shape[1], dtype=torch.bool)
        max_cnn_len = 0
        for i in range(self.L):
            factor = self._factor(i)
            ac = att_cache[i:i + 1, :, ::factor, :] if cache_t1 > 0 else None
            cc = None if cnn_cache is None or cnn_cache.numel() == 0 else cnn_cache[i]
            xs, new_att, new_cnn = self._layer_eff(i, xs, None, pos_emb, mask_pad, ac, cc, return_caches=True)
            if i in self._strides():
                mask_pad = mask_pad[:, :, ::2]
                pos_emb = pos_emb[:, ::2, :]
            new_att = new_att[:, :, next_cache_start // factor:, :]
            new_att = torch.repeat_interleave(new_att, factor, dim=2)
            new_cnn = new_cnn.unsqueeze(0)
            new_cnn = F.pad(new_cnn, (self.k - 1 - new_cnn.shape[3], 0))  # left-pad to cnn_module_kernel - 1 (:371-374)
            if i == 0:
                max_cnn_len = new_cnn.shape[3]
            r_att.append(new_att)  # not trimmed to the first layer's length (:380-382)
            r_cnn.append(new_cnn[:, :, :, -max_cnn_len:])
        xs = self._ln(xs, "encoder.after_norm")
        return xs, torch.cat(r_att, dim=0), torch.cat(r_cnn, dim=0)

    def get_encoder_out_chunk(self, speech, offset, required_cache_size, att_cache=None, cnn_cache=None):
        with torch.no_grad():
            xs, att_cache, cnn_cache = self.forward_chunk(speech, offset, required_cache_size, att_cache, cnn_cache)
            probs = torch.softmax(self.ctc_logits(xs), dim=2)
        return probs, att_cache, cnn_cache

    def encoder_forward(self, speech, speech_lengths, return_layers=False):
        # EfficientConformerEncoder.forward  efficient_conformer/encoder.py:212-264
        xs = torch.as_tensor(speech, dtype=self.dtype)
        lens = torch.as_tensor(speech_lengths, dtype=torch.int64)
        T = xs.shape[1]
        masks = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(1)
        xs = self._cmvn(xs)
        xs, pos_emb = self._embed(xs, 0)
        masks = masks[:, :, :-2:2][:, :, :-2:2]
        mask_pad = masks
        chunk_masks = masks & torch.ones(1, xs.shape[1], xs.shape[1], dtype=torch.bool)
        layers = [xs]
        for i in range(self.L):
            xs = self._layer_eff(i, xs, chunk_masks, pos_emb, mask_pad)
            if i in self._strides():
                masks = masks[:, :, ::2]
                chunk_masks = chunk_masks[:, ::2, ::2]
                mask_pad = masks
                pos_emb = pos_emb[:, ::2, :]
            layers.append(xs)
        xs = self._ln(xs, "encoder.after_norm")
        if return_layers:
            return xs, masks, layers
        return xs, masks
