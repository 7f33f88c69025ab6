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
                 max_len=5000, dtype=torch.float32):
        sd = dict(sd)
        sd.setdefault("encoder.after_norm.weight", sd["encoder.preln.weight"])  # only used for self.d
        super().__init__(sd, attention_heads, num_blocks, cnn_module_kernel, True, max_len, dtype)
        self.reduce_idx = reduce_idx
        self.recover_idx = recover_idx

    def _embed_sq(self, x):
        # DepthwiseConv2DSubsampling4.forward  squeezeformer/subsampling.py:53-68 (dw_stride False -> groups=1)
        x = x.unsqueeze(1)
        x = F.relu(F.conv2d(x, self.p["encoder.embed.pw_conv.weight"], self.p["encoder.embed.pw_conv.bias"], stride=2))
        x = F.relu(F.conv2d(x, self.p["encoder.embed.dw_conv.weight"], self.p["encoder.embed.dw_conv.bias"], stride=2))
        b, c, t, f = x.shape
        x = x.permute(0, 2, 1, 3).reshape(b, t, c * f)
        x = x * math.sqrt(self.d)  # RelPositionalEncoding on the c*f-wide tensor (pos_emb not added)
        pos_emb = self.pe[:, 0:t]
        x = self._linear(x, "encoder.embed.input_proj.0")
        return x, pos_emb

    def _attention_sq(self, x, mask, pos_emb, prefix):
        # squeezeformer/attention.py:96-162 (adaptive scale :120-123, linear_pos WITH bias :28)
        x = self.p[prefix + ".ada_scale"].reshape(1, 1, -1) * x + self.p[prefix + ".ada_bias"].reshape(1, 1, -1)
        B, T, _ = x.shape
        h, dk = self.h, self.dk
        q = self._linear(x, prefix + ".linear_q").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        k = self._linear(x, prefix + ".linear_k").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        v = self._linear(x, prefix + ".linear_v").reshape(B, T, h, dk).permute(0, 2, 1, 3)
        p = self._linear(pos_emb, prefix + ".linear_pos").reshape(1, -1, h, dk).permute(0, 2, 1, 3)
        q_u = q + self.p[prefix + ".pos_bias_u"].unsqueeze(1)
        q_v = q + self.p[prefix + ".pos_bias_v"].unsqueeze(1)
        scores = (q_u @ k.transpose(-1, -2) + q_v @ p.transpose(-1, -2)) / math.sqrt(dk)
        m = (mask.unsqueeze(1) == 0)[:, :, :, :scores.shape[-1]]
        scores = scores.masked_fill(m, -float("inf"))
        attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
        ctx = (attn @ v).permute(0, 2, 1, 3).reshape(B, T, h * dk)
        return self._linear(ctx, prefix + ".linear_out")

    def _ffn_sq(self, x, prefix):
        # squeezeformer/positionwise.py:55-65
        x = self.p[prefix + ".ada_scale"].reshape(1, 1, -1) * x + self.p[prefix + ".ada_bias"].reshape(1, 1, -1)
        return self._linear(self._swish(self._linear(x, prefix + ".w_1")), prefix + ".w_2")

    def _conv_sq(self, x, mask_pad, prefix):
        # squeezeformer/convolution.py:102-163 ; mask_pad True = valid here (fill where ~mask_pad)
        x = self.p[prefix + ".ada_scale"].reshape(1, 1, -1) * x + self.p[prefix + ".ada_bias"].reshape(1, 1, -1)
        x = x.transpose(1, 2)
        x = x.masked_fill(~mask_pad, 0.0)
        x = F.pad(x, (self.lorder, 0), "constant", 0.0)
        x = F.conv1d(x, self.p[prefix + ".pointwise_conv1.weight"], self.p[prefix + ".pointwise_conv1.bias"])
        x = F.glu(x, dim=1)
        if self.trace is not None:
            self.trace[prefix + ".glu"] = x.transpose(1, 2)
        x = F.conv1d(x, self.p[prefix + ".depthwise_conv.weight"], self.p[prefix + ".depthwise_conv.bias"],
                     groups=x.shape[1])
        x = x.transpose(1, 2)
        x = self._swish(self._ln(x, prefix + ".norm"))
        x = x.transpose(1, 2)
        x = F.conv1d(x, self.p[prefix + ".pointwise_conv2.weight"], self.p[prefix + ".pointwise_conv2.bias"])
        x = x.masked_fill(~mask_pad, 0.0)
        return x.transpose(1, 2)

    def _layer_sq(self, i, x, mask, pos_emb, mask_pad):
        # SqueezeformerEncoderLayer.forward  squeezeformer/encoder.py:435-506 (normalize_before=False)
        p = f"encoder.encoders.{i}"
        x = self._ln(x + self._attention_sq(x, mask, pos_emb, p + ".self_attn"), p + ".layer_norm1")
        x = self._ln(x + self._ffn_sq(x, p + ".ffn1"), p + ".layer_norm2")
        if self.trace is not None:
            self.trace[p + ".x2"] = x
        x = self._ln(x + self._conv_sq(x, mask_pad, p + ".conv_module"), p + ".layer_norm3")
        x = self._ln(x + self._ffn_sq(x, p + ".ffn2"), p + ".layer_norm4")
        return x

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
                # TimeReductionLayerStream.forward  time_reduction.py:183-206
                y = xs.transpose(1, 2).masked_fill(mask_pad == 0, 0.0)
                y = F.conv1d(y, self.p["encoder.time_reduction_layer.dw_conv.weight"],
                             self.p["encoder.time_reduction_layer.dw_conv.bias"], stride=2, groups=y.shape[1])
                y = F.conv1d(y, self.p["encoder.time_reduction_layer.pw_conv.weight"],
                             self.p["encoder.time_reduction_layer.pw_conv.bias"])
                xs = y.transpose(1, 2)
                chunk_masks = chunk_masks[:, ::2, ::2]
                mask_pad = mask_pad[:, :, ::2]
                Lr, Tr = mask_pad.shape[-1], xs.shape[1]
                if Lr - Tr < 0:
                    xs = xs[:, :Lr - Tr, :]
                elif Lr - Tr > 0:
                    xs = torch.cat([xs, torch.zeros(xs.shape[0], Lr - Tr, xs.shape[2], dtype=xs.dtype)], dim=1)
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
