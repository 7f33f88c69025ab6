"""evaluate() throughput with the decode of batch i overlapped with the encoder of batch i+1 (two HIP streams) against the
serial loop: Conformer 12 x 256, 32 x 10 s utterances per batch, PPASR's default beam search (beam 300, cutoff 0.99 / 40)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.decoders.beam_search_decoder import BeamSearchDecoder
from ppasr_amd.evaluate import evaluate
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.utils.synth import conformer_state_dict, synth_features, synth_vocabulary

V, L, B, NB = 4233, 12, 32, 6
vocab = synth_vocabulary(V)
model = ConformerModel(80, V, streaming=True, encoder_conf=dict(output_size=256, attention_heads=4, linear_units=2048,
                                                              num_blocks=L, cnn_module_kernel=15),
                       state_dict=conformer_state_dict(vocab_size=V, num_blocks=L, seed=1))
rng = np.random.Generator(np.random.PCG64(0))
batches = []
for i in range(NB):
    x, lens = synth_features(B, 1000, seed=i)
    batches.append((torch.from_numpy(x).cuda(), rng.integers(2, V - 1, size=(B, 40)).astype(np.int64), torch.from_numpy(lens).cuda(), None))
for beam in (10, 300):
    bsd = BeamSearchDecoder(0.0, 0.0, beam, 0.99, 40, vocab)
    res = {"beam": beam}
    for label, ov in (("serial", False), ("overlapped", True)):
        evaluate(model, batches[:2], vocab, decoder="ctc_beam_search", beam_search_decoder=bsd, overlap_decode=ov)
        torch.cuda.synchronize()
        t = time.perf_counter()
        evaluate(model, batches, vocab, decoder="ctc_beam_search", beam_search_decoder=bsd, overlap_decode=ov)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        res[label + "_ms_per_batch"] = round(dt / NB * 1e3, 2)
        res[label + "_audio_s_per_s"] = round(NB * B * 10 / dt)
    print(json.dumps(res), flush=True)
