"""DeepSpeech2 (5 x 1024 LSTM) encoder + greedy throughput over batch sizes, streaming (unidirectional) and not."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decode_ids
from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
from ppasr_amd.utils.synth import deepspeech2_state_dict, synth_features

V = 4233
for streaming in (True, False):
    sd = deepspeech2_state_dict(vocab_size=V, streaming=streaming, seed=1)
    m = DeepSpeech2Model(80, V, streaming=streaming, encoder_conf=dict(num_rnn_layers=5, rnn_size=1024), state_dict=sd)
    for B in (1, 8, 32):
        x, lens = synth_features(B, 498, seed=B)
        x, lens = torch.from_numpy(x).cuda(), torch.from_numpy(lens).cuda()
        for _ in range(2):
            greedy_decode_ids(m.get_encoder_out(x, lens))
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 3
        for _ in range(n):
            greedy_decode_ids(m.get_encoder_out(x, lens))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        print(json.dumps({"streaming": streaming, "B": B, "ms": round(dt * 1e3, 2), "audio_s_per_s": round(B * 4.98 / dt)}), flush=True)
