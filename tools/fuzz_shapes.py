"""Ad-hoc sweep of the model shapes outside the shipped YAMLs (input_layer conv2d6 / conv2d8, output_size 512 with 8 heads,
conv-module BatchNorm, non-causal variants) on random ragged batches: logits vs the torch-CPU oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.conformer_oracle import ConformerOracle  # noqa: E402
from ppasr_amd.model_utils.conformer.model import ConformerModel  # noqa: E402
from ppasr_amd.utils.synth import conformer_state_dict, synth_features  # noqa: E402

rng = np.random.Generator(np.random.PCG64(777))
bad = n = 0
V, L = 83, 2
for kw in (dict(input_layer="conv2d6"), dict(input_layer="conv2d8"), dict(output_size=512, attention_heads=8),
           dict(input_layer="conv2d6", cnn_module_norm="batch_norm"), dict(output_size=512, attention_heads=8, cnn_module_norm="batch_norm")):
    for streaming in (True, False):
        sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=int(rng.integers(1 << 30)), perturb_norm=True, **kw)
        conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
        conf.update(kw)
        model = ConformerModel(80, V, streaming=streaming, encoder_conf=conf, state_dict=sd, device="cuda:0")
        oracle = ConformerOracle(sd, num_blocks=L, causal=streaming, attention_heads=conf["attention_heads"])
        tmin = {"conv2d6": 11, "conv2d8": 15}.get(kw.get("input_layer"), 7)
        for case in range(8):
            B = int(rng.integers(1, 9))
            T = int(rng.integers(tmin, 900))
            lens = [int(v) for v in rng.integers(0, T + 1, size=B)]
            lens[0] = T
            x, la = synth_features(B, T, lens=lens, seed=case)
            _, logits = model.get_encoder_out(x, la, return_logits=True)
            _, ref = oracle.get_encoder_out(x, la, return_logits=True)
            torch.cuda.synchronize()
            err = float((logits.cpu() - ref).abs().max() / ref.abs().max())
            n += 1
            if not (err < 1e-4) or not bool(torch.isfinite(logits).all()):
                bad += 1
                print("FAIL", kw, streaming, B, T, lens, err)
        del model
print("fuzz_shapes done:", n, "cases,", bad, "problems")
