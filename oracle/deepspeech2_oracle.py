"""ORACLE (test infrastructure, not product code) -- PyTorch-CPU restatement of PPASR's DeepSpeech2
encoder + CTC head (ppasr/model_utils/deepspeech2/{model,encoder,conv}.py).  PARITY UNPINNED.

paddle.nn.LSTM semantics (from the Paddle 2.5 sources, not in the reference tree):
gate order i, f, g, o in the 4H rows; weight_ih [4H,in], weight_hh [4H,H], two biases; with
``sequence_length`` the outputs at t >= len are zero and the final states are those of the last valid
step; the reverse direction runs over the valid part only."""
import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, name, dtype):
    return torch.from_numpy(np.ascontiguousarray(sd[name])).to(dtype)


class DeepSpeech2Oracle:
    def __init__(self, sd, num_rnn_layers=5, rnn_size=1024, streaming=True, dtype=torch.float32, use_gru=False):
        self.p = {k: _t(sd, k, dtype) for k in sd}
        self.use_gru = use_gru  # nn.GRU instead of nn.LSTM (deepspeech2/encoder.py:36-42)
        self.L, self.H = num_rnn_layers, rnn_size
        self.dirs = 1 if streaming else 2  # rnn_direction 'forward' / 'bidirect' (deepspeech2/model.py:40)
        self.dtype = dtype

    def _lstm_dir(self, x, lens, prefix, sfx, h0, c0, reverse):
        B, T, _ = x.shape
        H = self.H
        w_ih, w_hh = self.p[prefix + "weight_ih" + sfx], self.p[prefix + "weight_hh" + sfx]
        b = self.p[prefix + "bias_ih" + sfx] + self.p[prefix + "bias_hh" + sfx]
        out = torch.zeros(B, T, H, dtype=x.dtype)
        hT, cT = h0.clone(), c0.clone()
        for bi in range(B):
            n = int(lens[bi])
            h, c = h0[bi], c0[bi]
            order = range(n - 1, -1, -1) if reverse else range(n)
            for t in order:
                if self.use_gru:
                    # paddle GRUCell: r, z, c rows; c = tanh(x_c + r * (W_hc h + b_hc)); h = (h - c) * z + c
                    xg = w_ih @ x[bi, t] + self.p[prefix + "bias_ih" + sfx]
                    hg = w_hh @ h + self.p[prefix + "bias_hh" + sfx]
                    r = torch.sigmoid(xg[:H] + hg[:H])
                    z = torch.sigmoid(xg[H:2 * H] + hg[H:2 * H])
                    cand = torch.tanh(xg[2 * H:] + r * hg[2 * H:])
                    h = (h - cand) * z + cand
                    out[bi, t] = h
                    continue
                g = w_ih @ x[bi, t] + w_hh @ h + b
                i, f, gg, o = g[:H], g[H:2 * H], g[2 * H:3 * H], g[3 * H:]
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h = torch.sigmoid(o) * torch.tanh(c)
                out[bi, t] = h
            hT[bi], cT[bi] = h, c
        return out, hT, cT

    def forward(self, speech, speech_lengths, init_h=None, init_c=None):
        """CRNNEncoder.forward (deepspeech2/encoder.py:61-104) + decoder.softmax (model.py:62-72)
        -> (probs, out_lens, final_h_box, final_c_box)"""
        with torch.no_grad():
            x = torch.as_tensor(speech, dtype=self.dtype)
            lens = torch.as_tensor(speech_lengths, dtype=torch.int64)
            x = (x - self.p["encoder.global_cmvn.mean"]) * self.p["encoder.global_cmvn.istd"]
            # Conv2dSubsampling4Pure.forward  conv.py:16-21
            x = x.unsqueeze(1)
            x = F.relu(F.conv2d(x, self.p["encoder.conv.conv.0.weight"], self.p["encoder.conv.conv.0.bias"], stride=2))
            x = F.relu(F.conv2d(x, self.p["encoder.conv.conv.2.weight"], self.p["encoder.conv.conv.2.bias"], stride=2))
            b, c, t, f = x.shape
            x = x.permute(0, 2, 1, 3).reshape(b, t, c * f)
            x_lens = ((lens - 1) // 2 - 1) // 2
            B, H, D = b, self.H, self.dirs
            hs, cs = [], []
            for l in range(self.L):
                outs = []
                for d in range(D):
                    idx = l * D + d
                    h0 = torch.zeros(B, H, dtype=self.dtype) if init_h is None else torch.as_tensor(init_h)[idx]
                    c0 = torch.zeros(B, H, dtype=self.dtype) if init_c is None else torch.as_tensor(init_c)[idx]
                    o, hT, cT = self._lstm_dir(x, x_lens, f"encoder.rnn.{l}.", "_l0" if d == 0 else "_l0_reverse",
                                               h0, c0, reverse=(d == 1))
                    outs.append(o)
                    hs.append(hT)
                    cs.append(cT)
                x = torch.cat(outs, dim=-1)
                x = F.layer_norm(x, (x.shape[-1],), self.p[f"encoder.layernorm_list.{l}.weight"],
                                 self.p[f"encoder.layernorm_list.{l}.bias"], 1e-5)
            logits = x @ self.p["decoder.ctc_lo.weight"] + self.p["decoder.ctc_lo.bias"]
            # GRU: the c box is handed through unchanged (encoder.py:95-97)
            c_box = torch.stack(cs) if not self.use_gru else (None if init_c is None else torch.as_tensor(init_c))
            return torch.softmax(logits, dim=2), x_lens, torch.stack(hs), c_box
