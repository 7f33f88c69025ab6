"""Streaming serving throughput: N independent Conformer sessions (device-resident caches), each fed 0.64 s chunks, on N HIP
streams from one host thread.  Prints the aggregate audio-seconds/s and the single-session chunk latency."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.utils.synth import DEFAULT_VOCAB_SIZE, conformer_state_dict, synth_features

V = DEFAULT_VOCAB_SIZE
conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12, cnn_module_kernel=15)
sd = conformer_state_dict(vocab_size=V, num_blocks=12, seed=1234)
model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
x, _ = synth_features(1, 67, seed=5)
chunk = torch.from_numpy(x).cuda()
n_chunks = 10
for n_sessions in (1, 8, 32, 64):
    sessions = [model.new_stream() for _ in range(n_sessions)]
    streams = [torch.cuda.Stream() for _ in range(n_sessions)]
    for rep in range(2):  # first repetition = warm-up
        for s in sessions:
            s.reset()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n_chunks):
            for s, st in zip(sessions, streams):
                with torch.cuda.stream(st):
                    s.encode_chunk(chunk, -16, want_probs=False, want_frames=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    audio = n_sessions * n_chunks * 0.64
    print(json.dumps({"sessions": n_sessions, "ms_per_chunk_round": round(dt / n_chunks * 1e3, 2),
                      "audio_s_per_s": round(audio / dt, 1)}), flush=True)

# ---- the same sessions advanced as ONE group call per chunk round ----
from ppasr_amd.model_utils.conformer.model import ConformerStreamGroup
for n_sessions in (8, 64, 256, 496):
    group = ConformerStreamGroup(model, n_sessions, max_frames=16 * (n_chunks * 2 + 2))
    batch = chunk.repeat(n_sessions, 1, 1).contiguous()
    ids = list(range(n_sessions))
    for rep in range(2):
        group.reset()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n_chunks):
            group.encode_chunks(ids, batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(json.dumps({"group_sessions": n_sessions, "ms_per_chunk_round": round(dt / n_chunks * 1e3, 2),
                      "audio_s_per_s": round(n_sessions * n_chunks * 0.64 / dt, 1)}), flush=True)
    del group

# ---- the opt-in fp16 x3 GEMM mode (ppasr_set_gemm_mode): the chunk's split-route kernels run their units on that route ----
model.set_gemm_mode("f16x3")
for n_sessions in (1, 8):
    sessions = [model.new_stream() for _ in range(n_sessions)]
    streams = [torch.cuda.Stream() for _ in range(n_sessions)]
    for rep in range(2):
        for s in sessions:
            s.reset()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n_chunks):
            for s, st in zip(sessions, streams):
                with torch.cuda.stream(st):
                    s.encode_chunk(chunk, -16, want_probs=False, want_frames=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(json.dumps({"gemm": "f16x3", "sessions": n_sessions, "ms_per_chunk_round": round(dt / n_chunks * 1e3, 2),
                      "audio_s_per_s": round(n_sessions * n_chunks * 0.64 / dt, 1)}), flush=True)
for n_sessions in (8, 64):
    group = ConformerStreamGroup(model, n_sessions, max_frames=16 * (n_chunks * 2 + 2))
    batch = chunk.repeat(n_sessions, 1, 1).contiguous()
    ids = list(range(n_sessions))
    for rep in range(2):
        group.reset()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n_chunks):
            group.encode_chunks(ids, batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(json.dumps({"gemm": "f16x3", "group_sessions": n_sessions, "ms_per_chunk_round": round(dt / n_chunks * 1e3, 2),
                      "audio_s_per_s": round(n_sessions * n_chunks * 0.64 / dt, 1)}), flush=True)
    del group
