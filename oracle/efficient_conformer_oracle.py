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
                 group_layer_idx=(0, 1, 2, 3), group_size=3, max_len=5000, dtype=torch.float32):
        super().__init__(sd, attention_heads, num_blocks, cnn_module_kernel, True, max_len, dtype)
        self.stride_layer_idx = stride_layer_idx
        self.group_layer_idx = tuple(group_layer_idx or ())
        self.group_size = group_size

    def _kernel(self, i):
        # cnn_module_kernels: halves after each stride layer (encoder.py:123-128, stride_kernel=True)
        if self.stride_layer_idx is not None and i > self.stride_layer_idx:
            return self.k // 2
        return self.k

    def _grouped_attention(self, x, mask, pos_emb, prefix):
        # GroupedRelPositionMultiHeadedAttention.forward  efficient_conformer/attention.py:128-193
        B, T, _ = x.shape
        h, dk, g = self.h, self.dk, self.group_size
        q = self._linear(x, prefix + ".linear_q").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        k = self._linear(x, prefix + ".linear_k").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        v = self._linear(x, prefix + ".linear_v").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        p = self._linear(pos_emb, prefix + ".linear_pos")  # with bias (:31)
        # pad4group :40-79
        pad_t = (g - T % g) % g
        q = F.pad(q, (0, 0, 0, pad_t))
        k = F.pad(k, (0, 0, 0, pad_t))
        v = F.pad(v, (0, 0, 0, pad_t))
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
        m = (mask.unsqueeze(1) == 0)[:, :, :, :scores.shape[-1]]
        scores = scores.masked_fill(m, -float("inf"))
        attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
        ctx = (attn @ v).permute(0, 2, 1, 3).reshape(B, -1, h * dk)
        ctx = ctx[:, :ctx.shape[1] - pad_t]
        if self.trace is not None:
            self.trace[prefix + ".ctx"] = ctx
        return self._linear(ctx, prefix + ".linear_out")

    def _conv_eff(self, x, mask_pad, prefix, ksize, stride):
        # efficient_conformer/convolution.py:80-138 ; mask_pad True = valid
        lorder = ksize - 1
        x = x.transpose(1, 2).masked_fill(~mask_pad, 0.0)
        x = F.pad(x, (lorder, 0), "constant", 0.0)
        x = F.conv1d(x, self.p[prefix + ".pointwise_conv1.weight"], self.p[prefix + ".pointwise_conv1.bias"])
        x = F.glu(x, dim=1)
        x = F.conv1d(x, self.p[prefix + ".depthwise_conv.weight"], self.p[prefix + ".depthwise_conv.bias"],
                     stride=stride, groups=x.shape[1])
        x = x.transpose(1, 2)
        x = self._swish(self._ln(x, prefix + ".norm"))
        x = x.transpose(1, 2)
        x = F.conv1d(x, self.p[prefix + ".pointwise_conv2.weight"], self.p[prefix + ".pointwise_conv2.bias"])
        if mask_pad.shape[2] != x.shape[2]:
            mask_pad = mask_pad[:, :, ::stride]
        x = x.masked_fill(~mask_pad, 0.0)
        return x.transpose(1, 2)

    def _layer_eff(self, i, x, mask, pos_emb, mask_pad):
        # ConformerEncoderLayer / StrideConformerEncoderLayer (efficient_conformer/encoder.py:455-548)
        p = f"encoder.encoders.{i}"
        x = x + 0.5 * self._ffn(self._ln(x, p + ".norm_ff_macaron"), p + ".feed_forward_macaron")
        xn = self._ln(x, p + ".norm_mha")
        if i in self.group_layer_idx:
            x = x + self._grouped_attention(xn, mask, pos_emb, p + ".self_attn")
        else:
            x_att, _ = self._attention(xn, mask, pos_emb, None, p + ".self_attn")
            x = x + x_att
        residual = x
        stride = 2 if (self.stride_layer_idx is not None and i == self.stride_layer_idx) else 1
        y = self._conv_eff(self._ln(x, p + ".norm_conv"), mask_pad, p + ".conv_module", self._kernel(i), stride)
        if stride > 1:
            # paddle.nn.AvgPool1D(2, 2, padding=0, ceil_mode=True), exclusive (encoder.py:171-172)
            residual = F.avg_pool1d(residual.transpose(1, 2), kernel_size=2, stride=2, padding=0, ceil_mode=True,
                                    count_include_pad=False).transpose(1, 2)
        x = residual + y
        x = x + 0.5 * self._ffn(self._ln(x, p + ".norm_ff"), p + ".feed_forward")
        return self._ln(x, p + ".norm_final")

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
            if self.stride_layer_idx is not None and i == self.stride_layer_idx:
                masks = masks[:, :, ::2]
                chunk_masks = chunk_masks[:, ::2, ::2]
                mask_pad = masks
                pos_emb = pos_emb[:, ::2, :]
            layers.append(xs)
        xs = self._ln(xs, "encoder.after_norm")
        if return_layers:
            return xs, masks, layers
        return xs, masks
