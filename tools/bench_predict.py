"""End-to-end latency of the public surface on one utterance: PPASRPredictor.predict (wav in, text out: fbank -> encoder ->
decode) and predict_stream (0.5 s PCM chunks), 12-block Conformer, V = 4233, greedy and beam search (PPASR's default
beam 300)."""
import cProfile
import json
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.predict import PPASRPredictor
from ppasr_amd.utils.synth import conformer_state_dict, synth_vocabulary

V, L = 4233, 12
vocab = synth_vocabulary(V)
sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=1)


def cfg(decoder, beam):
    enc = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
    return dict(encoder_conf=enc, preprocess_conf=dict(feature_method="fbank", n_mels=80, sample_rate=16000,
                                                        use_dB_normalization=True, target_dB=-20),
                ctc_beam_search_decoder_conf=dict(alpha=2.2, beta=4.3, beam_size=beam, num_processes=10, cutoff_prob=0.99,
                                                  cutoff_top_n=40, language_model_path=None),
                use_model="conformer", streaming=True, decoder=decoder, metrics_type="cer")


rng = np.random.Generator(np.random.PCG64(0))
t = np.arange(16000 * 10) / 16000.0
wav = (0.1 * np.sin(2 * np.pi * 220 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.02 * rng.standard_normal(t.shape)).astype(np.float32)
pcm = (np.clip(wav, -1, 1) * 32767).astype(np.int16).tobytes()
for decoder, beam in (("ctc_greedy", 10), ("ctc_beam_search", 300)):
    p = PPASRPredictor(configs=cfg(decoder, beam), state_dict=sd, vocab_list=vocab, warmup=True)
    for _ in range(3):
        p.predict(audio_data=wav)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        p.predict(audio_data=wav)
    dt = (time.perf_counter() - t0) / n
    # streaming: 0.5 s chunks
    step = 16000
    p.reset_stream()
    lat = []
    for i in range(0, len(pcm), step):
        t1 = time.perf_counter()
        p.predict_stream(audio_data=pcm[i:i + step], is_end=(i + step >= len(pcm)))
        lat.append(time.perf_counter() - t1)
    p.reset_stream()
    print(json.dumps({"decoder": decoder, "beam": beam, "predict_10s_ms": round(dt * 1e3, 2),
                      "predict_stream_0.5s_chunk_ms_mean": round(float(np.mean(lat[1:])) * 1e3, 2),
                      "predict_stream_0.5s_chunk_ms_max": round(float(np.max(lat[1:])) * 1e3, 2)}), flush=True)
    if "--profile" in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            p.predict(audio_data=wav)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
