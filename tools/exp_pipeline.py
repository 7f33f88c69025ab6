"""Experiment: K steps of the cfg2 workload issued on ONE stream vs alternately on TWO streams (each stream's steps are
serial, the two streams overlap: ramps / tails / the HBM-bound conv1 of one batch fill the other's gaps)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.utils.synth import conformer_state_dict, synth_features
V, L = 4233, 12
conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
m = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=conformer_state_dict(vocab_size=V, num_blocks=L, seed=1234))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x, l = synth_features(B, 1000, seed=20440)
x, l = torch.from_numpy(x).cuda(), torch.from_numpy(l).cuda()
K = 200
for ns in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for s in streams:
        with torch.cuda.stream(s):
            m.encode_greedy(x, l)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(K):
        with torch.cuda.stream(streams[i % ns]):
            out = m.encode_greedy(x, l)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    print(f"B={B} streams={ns}: {dt*1e3:.3f} ms/step  {B*10/dt:.0f} audio-s/s", flush=True)
