import os, sys, time, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.utils.synth import conformer_state_dict, synth_features
V, L = 4233, 12
m = ConformerModel(80, V, streaming=True, encoder_conf=dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15), state_dict=conformer_state_dict(vocab_size=V, num_blocks=L, seed=1))
out = {}
for B in (1, 2, 4, 8, 16):
    x, lens = synth_features(B, 1000, seed=2)
    x = torch.from_numpy(x).cuda(); lens = torch.from_numpy(lens).cuda()
    for _ in range(3): m.encode_greedy(x, lens)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): m.encode_greedy(x, lens)
    torch.cuda.synchronize(); out[B] = round((time.perf_counter() - t) / 10 * 1e3, 3)
print(os.environ.get("PPASR_ATTN_FUSE_MIN_BLOCKS"), json.dumps(out))
