"""Per-kernel time of the generic-width route (output_size 512 / 8 heads, 12 blocks, 32 x 10 s, V = 4233, greedy):
dispatch-attached HIP events (ppasr_kprof_*), scaled to the un-instrumented step time.  One JSON line."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd._lib import kernel_profile
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.utils.synth import conformer_state_dict, synth_features

V = 4233
D = int(os.environ.get("GEN_D", "512"))
x, lens = synth_features(32, 1000, seed=20440)
x, lens = torch.from_numpy(x).cuda(), torch.from_numpy(lens).cuda()
kw = dict(output_size=D, attention_heads=D // 64)
sd = conformer_state_dict(vocab_size=V, num_blocks=12, seed=1234, **kw)
conf = dict(output_size=D, attention_heads=D // 64, linear_units=2048, num_blocks=12, cnn_module_kernel=15)
m = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd)
step = lambda: m.encode_greedy(x, lens)
for _ in range(2):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 10
with kernel_profile() as kp:
    step()
    torch.cuda.synchronize()
tot = sum(ms for ms, _ in kp.kernels.values())
sc = dt * 1e3 / tot
rows = {n: {"launches": c, "ms": round(ms * sc, 3), "avg_us": round(ms * sc / c * 1e3, 1)}
        for n, (ms, c) in sorted(kp.kernels.items(), key=lambda kv: -kv[1][0])}
print(json.dumps({"shape": f"output_size {D}", "ms": round(dt * 1e3, 2), "audio_s_per_s": round(320 / dt), "kernels": rows}, indent=1))
