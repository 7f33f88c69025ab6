"""cfg2's step (32 x 10 s, Conformer, greedy) N times in the given GEMM mode -- for rocprofv3 --kernel-trace --stats."""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.utils.synth import conformer_state_dict, synth_features

mode, n = sys.argv[1], int(sys.argv[2])
V, L = 4233, 12
sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=1234)
conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
m = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
x, la = synth_features(32, 1000, seed=20240 + 200)
x, la = torch.from_numpy(x).cuda(), torch.from_numpy(la).cuda()
m.set_gemm_mode(mode)
for _ in range(n):
    m.encode_greedy(x, la)
torch.cuda.synchronize()
