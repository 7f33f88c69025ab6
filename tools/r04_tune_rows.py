"""Squeezeformer encoder + greedy over batch sizes: fused 32-row kernels / split route (ppasr_set_ffn_split auto) /
16-row kernels (ppasr_set_row_block 16) / the default route selection -- which under-filled-launch route wins where."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
from ppasr_amd.utils.synth import squeezeformer_state_dict, synth_features

V, L = 4233, 12
m = SqueezeformerModel(80, V, streaming=True, encoder_conf=dict(
    encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=5, recover_idx=11,
    feed_forward_expansion_factor=8, cnn_module_kernel=31), state_dict=squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=1))


def timeit(x, lens, n=6):
    for _ in range(2):
        m.encode_greedy(x, lens)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        m.encode_greedy(x, lens)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for B in (1, 2, 4, 6, 8, 12, 16, 24, 32, 48):
    x, lens = synth_features(B, 1000, seed=B)
    x, lens = torch.from_numpy(x).cuda(), torch.from_numpy(lens).cuda()
    res = {"B": B, "blocks32_full": (B * 249 + 31) // 32, "blocks32_half": (B * 125 + 31) // 32}
    for label, rows, split in (("fused32", 32, 0), ("split", 32, -1), ("rows16", 16, 0), ("default", -1, -1)):
        m.set_row_block(rows)
        m.set_ffn_split(split)
        res[label + "_ms"] = round(timeit(x, lens), 3)
    m.set_row_block(-1)
    m.set_ffn_split(-1)
    print(json.dumps(res), flush=True)
