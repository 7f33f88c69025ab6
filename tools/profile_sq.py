"""rocprofv3 helper: a few Squeezeformer / Efficient-Conformer steps at the bench shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
from ppasr_amd.utils.synth import squeezeformer_state_dict, synth_features
V, L = 4233, 12
sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=1)
conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=5, recover_idx=11,
            feed_forward_expansion_factor=8, cnn_module_kernel=31)
m = SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd)
x, lens = synth_features(32, 1000, seed=2)
x = torch.from_numpy(x).cuda(); lens = torch.from_numpy(lens).cuda()
for _ in range(3):
    m.encode_greedy(x, lens)
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(5):
    m.encode_greedy(x, lens)
torch.cuda.synchronize()
print("squeezeformer ms/step", (time.perf_counter() - t) / 5 * 1e3)
