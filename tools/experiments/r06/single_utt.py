"""Encoder latency of ONE utterance (B = 1) of 5 / 10 / 20 s: the split route of under-filled launches (A/B of library knobs in
separate processes)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.utils.synth import DEFAULT_VOCAB_SIZE, conformer_state_dict, synth_features

V = DEFAULT_VOCAB_SIZE
conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12, cnn_module_kernel=15)
sd = conformer_state_dict(vocab_size=V, num_blocks=12, seed=1234)
model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
out = {"tag": os.environ.get("TAG", "")}
for T in [int(v) for v in os.environ.get("TS", "500,1000,2000").split(",")]:
    x, la = synth_features(1, T, seed=5)
    xd = torch.from_numpy(x).cuda()
    lad = torch.as_tensor(la).cuda()
    best = 1e9
    for rep in range(6):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            model.get_encoder_out(xd, lad)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 20 * 1e3
        if rep:
            best = min(best, dt)
    out[f"T{T}_ms"] = round(best, 3)
print(json.dumps(out), flush=True)
