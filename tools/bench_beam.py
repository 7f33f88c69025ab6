import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
rng = np.random.Generator(np.random.PCG64(0))
B, T, V = 32, 249, 4233
logits = rng.standard_normal((B, T, V)).astype(np.float32) * 3
p = torch.softmax(torch.from_numpy(logits), -1).cuda()
for beam in (10, 100, 300):
    beam_search_ids(p, beam, 0.99, 40, 0); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): beam_search_ids(p, beam, 0.99, 40, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"beam {beam}: {dt*1e3:.2f} ms per batch of {B} x {T} frames = {dt/T*1e6:.1f} us/frame")
