"""Timings of the model shapes outside the shipped YAMLs (one GPU, 12 blocks, V = 4233, 32 x 10 s, greedy):
  output_size 512 / 8 heads on the generic-width route, input_layer conv2d6 / conv2d8, cnn_module_norm batch_norm.
Prints one JSON line per shape."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.utils.synth import conformer_state_dict, synth_features
V = 4233


def timeit(fn, steps=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps


x, lens = synth_features(32, 1000, seed=20440)
x, lens = torch.from_numpy(x).cuda(), torch.from_numpy(lens).cuda()
for name, kw in (("output_size 256 / 4 heads (the bench shape)", {}),
                 ("cnn_module_norm batch_norm", dict(cnn_module_norm="batch_norm")),
                 ("input_layer conv2d6", dict(input_layer="conv2d6")),
                 ("input_layer conv2d8", dict(input_layer="conv2d8")),
                 ("output_size 512 / 8 heads (general layer route)", dict(output_size=512, attention_heads=8))):
    sd = conformer_state_dict(vocab_size=V, num_blocks=12, seed=1234, **kw)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12, cnn_module_kernel=15)
    conf.update(kw)
    m = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd)
    dt = timeit(lambda: m.encode_greedy(x, lens), 10, 2)
    print(json.dumps({"shape": name, "ms": round(dt * 1e3, 2), "audio_s_per_s": round(320 / dt)}), flush=True)
    del m
