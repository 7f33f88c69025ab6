"""Run a few hot-path steps of the bench workload (for rocprofv3):
   rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python tools/profile_step.py --steps 5"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.model_utils.conformer.model import ConformerModel  # noqa: E402
from ppasr_amd.utils.synth import DEFAULT_VOCAB_SIZE, conformer_state_dict, synth_features  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--blocks", type=int, default=12)
ap.add_argument("--ffn", type=int, default=2048)
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()
V, L = DEFAULT_VOCAB_SIZE, args.blocks
conf = dict(output_size=256, attention_heads=4, linear_units=args.ffn, num_blocks=L, cnn_module_kernel=15)
sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=1234, linear_units=args.ffn)
model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
x, lens = synth_features(args.batch, args.frames, seed=20440)
x = torch.from_numpy(x).cuda()
lens = torch.from_numpy(lens).cuda()
for _ in range(args.steps):
    model.encode_greedy(x, lens)
torch.cuda.synchronize()
if args.profile:
    model.profile_kernels(True)
    model.encode_greedy(x, lens)
    for k, (ms, n) in model.read_kernel_profile().items():
        print(f"{k:24s} n={n:3d} avg_us={1e3 * ms / max(n, 1):8.1f}")
print("done")
