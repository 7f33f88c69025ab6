import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
rng = np.random.Generator(np.random.PCG64(0))
B, T, V = 32, 249, 4233
p = torch.softmax(torch.from_numpy(rng.standard_normal((B, T, V)).astype(np.float32) * 3), -1).cuda()
for beam in [int(a) for a in sys.argv[1:]] or [10, 300]:
    beam_search_ids(p, beam, 0.99, 40, 0); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): beam_search_ids(p, beam, 0.99, 40, 0)
    torch.cuda.synchronize()
    print(f"beam {beam}: {(time.perf_counter()-t)/3/T*1e6:.2f} us/frame", flush=True)
