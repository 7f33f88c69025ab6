"""One 0.64 s chunk of 1 / 32 independent streaming sessions, best of 5 rounds of 40 chunks (A/B of library knobs: run it
in separate processes, alternating)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.utils.synth import DEFAULT_VOCAB_SIZE, conformer_state_dict, synth_features

V = DEFAULT_VOCAB_SIZE
conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12, cnn_module_kernel=15)
sd = conformer_state_dict(vocab_size=V, num_blocks=12, seed=1234)
model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
x, _ = synth_features(1, 67, seed=5)
chunk = torch.from_numpy(x).cuda()
n_chunks = 40
out = {"tag": os.environ.get("TAG", "")}
for n_sessions in [int(v) for v in os.environ.get("SESS", "1,32").split(",")]:
    sessions = [model.new_stream() for _ in range(n_sessions)]
    streams = [torch.cuda.Stream() for _ in range(n_sessions)]
    best = 1e9
    for rep in range(6):
        for s in sessions:
            s.reset()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n_chunks):
            for s, st in zip(sessions, streams):
                with torch.cuda.stream(st):
                    s.encode_chunk(chunk, -16, want_probs=False, want_frames=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        if rep:
            best = min(best, dt / n_chunks * 1e3)
    out[f"sessions_{n_sessions}_ms"] = round(best, 3)
print(json.dumps(out), flush=True)
