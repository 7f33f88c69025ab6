import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
rng = np.random.Generator(np.random.PCG64(0))
B, T, V = 32, 249, 4233
logits = rng.standard_normal((B, T, V)).astype(np.float32) * 3
p = torch.softmax(torch.from_numpy(logits), -1).cuda()
for beam, env in ((10, {}), (10, {"PPASR_BEAM_FAST": "0"}), (100, {}), (300, {})):
    os.environ.pop("PPASR_BEAM_FAST", None); os.environ.update(env)
    print("== beam", beam, env, flush=True)
    beam_search_ids(p, beam, 0.99, 40, 0); torch.cuda.synchronize()
