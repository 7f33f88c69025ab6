"""One streaming session, 0.64 s chunks: the split route (default for one row block) against the fused layer kernels
(ppasr_set_ffn_split(h, 0)): chunk latency, launches per chunk and the kernels of a chunk."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from ppasr_amd._lib import kernel_profile
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.utils.synth import DEFAULT_VOCAB_SIZE, conformer_state_dict, synth_features

V = DEFAULT_VOCAB_SIZE
conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12, cnn_module_kernel=15)
model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=conformer_state_dict(vocab_size=V, num_blocks=12, seed=1234),
                       device="cuda:0")
x, _ = synth_features(1, 67, seed=5)
chunk = torch.from_numpy(x).cuda()
for route in (-1, 0, 2, 4, "f16x3"):
    if route == "f16x3":  # the default (split) route with its units on the fp16 x3 route
        model.set_ffn_split(-1)
        model.set_gemm_mode("f16x3")
    else:
        model.set_ffn_split(route)
    s = model.new_stream()
    for rep in range(3):
        s.reset()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            s.encode_chunk(chunk, -16, want_probs=False, want_frames=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 20
    s.reset()
    s.encode_chunk(chunk, -16, want_probs=False, want_frames=True)
    with kernel_profile(max_entries=256) as kp:
        s.encode_chunk(chunk, -16, want_probs=False, want_frames=True)
    torch.cuda.synchronize()
    n = sum(c for _ms, c in kp.kernels.values())
    top = sorted(kp.kernels.items(), key=lambda kv: -kv[1][0])[:6]
    print(json.dumps({"ffn_split": route, "ms_per_chunk": round(dt * 1e3, 3), "launches": n, "kernel_ms": round(sum(ms for ms, _ in kp.kernels.values()), 3),
                      "top": [(k, round(ms, 3), c) for k, (ms, c) in top]}), flush=True)
