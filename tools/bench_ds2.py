"""DeepSpeech2 (5 x 1024) encoder + greedy: throughput and per-kernel roofline figures over the shapes of VERDICT r02 #6 --
B = 1 bidirectional (BASELINE configs[0]), B = 32 / 128 unidirectional (streaming model), LSTM and GRU.
One JSON line per shape; kernel durations from dispatch-attached HIP events (ppasr_kprof_*).

Roofline conventions (NOTES.md 7, DeepSpeech2): the recurrence kernels re-read the recurrent weights every time step, so
their algorithmic BYTES per launch are those weights (k_lstm_step / k_gru_step: one utterance per launch, matrix-vector,
HBM / Infinity-Cache bound); the batched kernels (k_lstm_step_mfma, k_lstm_wave) are priced both ways: algorithmic FLOPs
against the fp32-MFMA peak and weight bytes against HBM -- with 32 rows per weight byte they sit at the ridge."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd._lib import kernel_profile
from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decode_ids
from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
from ppasr_amd.utils.synth import deepspeech2_state_dict, synth_features

V, H, L, T = 4233, 1024, 5, 498
Tp = ((T - 1) // 2 - 1) // 2
PEAK_TF, PEAK_GBS = 157.3, 8000.0


def shapes():
    uni_only = "uni" in sys.argv[1:]  # (kernel-variant A/B runs: the unidirectional LSTM shapes only)
    if not uni_only:
        yield dict(streaming=False, use_gru=False, B=1)
        yield dict(streaming=False, use_gru=True, B=1)
    for gru in ((False,) if uni_only else (False, True)):
        for B in ((32, 64, 128) if uni_only else (32, 128)):
            yield dict(streaming=True, use_gru=gru, B=B)


for sh in shapes():
    streaming, gru, B = sh["streaming"], sh["use_gru"], sh["B"]
    dirs, gates = (1 if streaming else 2), (3 if gru else 4)
    sd = deepspeech2_state_dict(vocab_size=V, streaming=streaming, seed=1, use_gru=gru)
    m = DeepSpeech2Model(80, V, streaming=streaming, encoder_conf=dict(num_rnn_layers=L, rnn_size=H, use_gru=gru), state_dict=sd)
    x, lens = synth_features(B, T, seed=B)
    x, lens = torch.from_numpy(x).cuda(), torch.from_numpy(lens).cuda()
    step = lambda: greedy_decode_ids(m.get_encoder_out(x, lens))
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    n = 5
    t = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    with kernel_profile() as kp:
        step()
        torch.cuda.synchronize()
    tot = sum(ms for ms, _ in kp.kernels.values())
    scale = dt * 1e3 / tot
    rec_flops = L * Tp * B * dirs * 2 * H * gates * H                       # h W_hh^T of every step
    in_flops = sum(2 * k * gates * H * Tp * B * dirs for k in [608] + [dirs * H] * (L - 1))
    kernels = {}
    for name, (ms, cnt) in sorted(kp.kernels.items(), key=lambda kv: -kv[1][0]):
        e = {"launches": cnt, "ms_per_step": round(ms * scale, 4), "avg_us": round(ms * scale / cnt * 1e3, 2)}
        base = name.split("<")[0]
        if base in ("k_lstm_step", "k_gru_step"):
            byts = cnt * dirs * gates * H * H * 4   # every launch reads the recurrent weights of the layer's direction(s)
            e.update(bound="hbm", gbs=round(byts / (ms * scale * 1e-3) / 1e9, 1), frac=round(byts / (ms * scale * 1e-3) / 1e9 / PEAK_GBS, 4))
        elif base in ("k_lstm_step_mfma", "k_lstm_wave"):   # (k_lstm_step_mfma<true> = the GRU layers on the same tiles)
            fl = rec_flops + (in_flops - 2 * 608 * gates * H * Tp * B * dirs if base == "k_lstm_wave" else 0)
            # weights a launch streams: one layer's W_hh per direction (step kernel); W_hh of every layer + the folded input
            # projection of layers >= 1 (the (layer, time) wavefront: all layers are in flight in one launch)
            wbytes = cnt * gates * H * H * 4 * (dirs if base == "k_lstm_step_mfma" else 2 * L - 1)
            e.update(bound="mfma|hbm", tflops=round(fl / (ms * scale * 1e-3) / 1e12, 2), frac_mfma=round(fl / (ms * scale * 1e-3) / 1e12 / PEAK_TF, 4),
                     weight_gbs=round(wbytes / (ms * scale * 1e-3) / 1e9, 1), frac_hbm=round(wbytes / (ms * scale * 1e-3) / 1e9 / PEAK_GBS, 4))
        elif base == "k_gemm_stream":
            # input projections (all layers, or layer 0 only when the wavefront kernel folds the others) + CTC head
            folded = any(k.startswith("k_lstm_wave") for k in kp.kernels)
            fl = (2 * 608 * gates * H * Tp * B * dirs if folded else in_flops) + 2 * dirs * H * V * Tp * B
            e.update(bound="mfma", tflops=round(fl / (ms * scale * 1e-3) / 1e12, 2), frac=round(fl / (ms * scale * 1e-3) / 1e12 / PEAK_TF, 4))
        kernels[name] = e
    print(json.dumps({"model": "DeepSpeech2 5x1024 " + ("GRU" if gru else "LSTM") + (" unidirectional" if streaming else " bidirectional"),
                      "B": B, "frames": T, "ms": round(dt * 1e3, 3), "audio_s_per_s": round(B * T * 0.01 / dt, 1), "kernels": kernels}), flush=True)
    del m
