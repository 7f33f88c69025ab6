"""Debug: Squeezeformer 12 blocks on the cfg5 bucket [1,2,3] (lens 2388/2355/2205): where does the HIP path leave the oracle?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_cases as rc  # noqa: E402
from oracle.squeezeformer_oracle import SqueezeformerOracle  # noqa: E402
from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel  # noqa: E402

torch.set_num_threads(32)
case = rc.FULL["cfg5"]
sd = rc.state_dict(case)
x, lens = rc.features(case)
conf = rc.product_encoder_conf(case)
model = SqueezeformerModel(80, case["V"], streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
orc = SqueezeformerOracle(sd, num_blocks=12)


def cmp(name, idx, Tb, split=-1):
    model.set_ffn_split(split)
    ll = np.minimum(lens[idx], Tb)
    p, l = model.get_encoder_out(x[idx, :Tb], ll, return_logits=True)
    torch.cuda.synchronize()
    _, ref = orc.get_encoder_out(x[idx, :Tb], ll, return_logits=True)
    got, ref = l.cpu().numpy(), ref.numpy()
    for j, i in enumerate(idx):
        err = np.abs(got[j] - ref[j]).max(-1) / np.abs(ref[j]).max()
        bad = np.nonzero(err > 1e-3)[0]
        print(f"{name} split={split} utt {i} len {lens[i]} Tp {got.shape[1]}: max rel {err.max():.2e} bad frames {len(bad)}"
              f" first {bad[:6]} last {bad[-6:]}", flush=True)


cmp("bucket[1,2,3]", [1, 2, 3], 2388)
cmp("bucket[1,2,3]", [1, 2, 3], 2388, split=0)
cmp("utt1 alone", [1], 2388)
cmp("utt1 alone", [1], 2388, split=0)
cmp("utt0 alone", [0], 2975)
cmp("bucket[5..8]", [5, 6, 7, 8], 1780)
cmp("bucket[9,10]", [9, 10], 1554)
for T in (2388, 2048, 2052, 2056, 1800, 1200, 900):
    cmp(f"utt1 cut to {T}", [1], T)
