"""Encoder + greedy throughput of the three *former families over batch sizes (32 x 10 s is the bench shape), with the
fused layer kernels only ("fused") and with the default route selection ("auto": launches with <= 128 row blocks take
the split route, ppasr_set_ffn_split).  A kernel with <= 256 row blocks takes one round whatever their number, so
small batches and the half-rate layers of Squeezeformer / Efficient-Conformer leave CUs idle on the fused route."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.utils.synth import (conformer_state_dict, efficient_conformer_state_dict, squeezeformer_state_dict,
                                   synth_features)

V, L = 4233, 12


def models():
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    yield "conformer", ConformerModel(80, V, streaming=True, encoder_conf=dict(
        output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15),
        state_dict=conformer_state_dict(vocab_size=V, num_blocks=L, seed=1))
    yield "squeezeformer", SqueezeformerModel(80, V, streaming=True, encoder_conf=dict(
        encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=5, recover_idx=11,
        feed_forward_expansion_factor=8, cnn_module_kernel=31), state_dict=squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=1))
    yield "efficient_conformer", EfficientConformerModel(80, V, streaming=True, encoder_conf=dict(
        output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15, cnn_module_norm="layer_norm",
        efficient_conf=dict(stride_layer_idx=[3], stride=[2], group_layer_idx=[0, 1, 2, 3], group_size=3)),
        state_dict=efficient_conformer_state_dict(vocab_size=V, seed=1))


only = sys.argv[1:]  # optional: model names
for name, m in models():
    if only and name not in only:
        continue
    for B in (4, 8, 16, 32, 64, 128):
        x, lens = synth_features(B, 1000, seed=B)
        x, lens = torch.from_numpy(x).cuda(), torch.from_numpy(lens).cuda()
        res = {"model": name, "B": B}
        for label, mode in (("fused", 0), ("auto", -1)):  # always the fused kernels / split route for under-filled grids
            m.set_ffn_split(mode)
            for _ in range(2):
                m.encode_greedy(x, lens)
            torch.cuda.synchronize()
            t = time.perf_counter()
            n = 5
            for _ in range(n):
                m.encode_greedy(x, lens)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / n
            res[label + "_ms"] = round(dt * 1e3, 2)
            res[label + "_audio_s_per_s"] = round(B * 10 / dt)
        print(json.dumps(res), flush=True)
    del m
    torch.cuda.empty_cache()
