"""Which half of evaluate(overlap_decode=True) is not deterministic under concurrency: the encoder's output bits, or the
beam search of fixed log-probabilities while another stream keeps the chip busy?  (r06: a 2-in-10 flake of
tests/test_predictor_gpu.py::test_evaluate_overlapped_decode_gives_the_same_result[ctc_beam_search].)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests"))

from ppasr_amd.decoders.beam_search_decoder import beam_search_ids  # noqa: E402
from ppasr_amd.model_utils.conformer.model import ConformerModel  # noqa: E402
from ppasr_amd.utils.synth import conformer_state_dict, synth_features, synth_vocabulary  # noqa: E402

V = 120
reps = int(os.environ.get("REPS", 40))
beams = [int(v) for v in os.environ.get("BEAMS", "20").split(",")]
sd = conformer_state_dict(vocab_size=V, num_blocks=2, seed=19)
conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=2, cnn_module_kernel=15)
model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
rng = np.random.Generator(np.random.PCG64(3))
batches = []
for seed in range(5):
    lens = sorted((int(v) for v in rng.integers(60, 400, size=4)), reverse=True)
    x, la = synth_features(4, lens[0], lens=lens, seed=seed)
    batches.append((x, la))

cutoff = float(os.environ.get("CUTOFF", "0.99"))
topn = int(os.environ.get("TOPN", "40"))
scorer = None
if os.environ.get("SCORER", "0") == "1":  # a character-based 3-gram scorer over most of the vocabulary
    import tempfile
    from lm_util import write_synthetic_arpa
    from ppasr_amd.decoders.beam_search_decoder import Scorer
    vocab = synth_vocabulary(V)
    arpa = write_synthetic_arpa(os.path.join(tempfile.mkdtemp(), "lm.arpa"), vocab[2:100], order=3, seed=4)
    scorer = Scorer(2.2, 4.3, arpa, vocab)
print("cutoff", cutoff, "top_n", topn, "scorer", scorer is not None, flush=True)
quiet = [model.get_encoder_out(x, la).clone() for x, la in batches]
torch.cuda.synchronize()
side = torch.cuda.Stream()
big = torch.randn(4096, 4096, device="cuda:0")

# 1. encoder bits while beam kernels run on the main stream
bad_enc = 0
for r in range(reps if os.environ.get("ENC", "1") == "1" else 0):
    for i, (x, la) in enumerate(batches):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            o = model.get_encoder_out(x, la)
        beam_search_ids(quiet[(i + 1) % 5], 20, 0.99, 40, 0, nbest=1)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if not torch.equal(o, quiet[i]):
            bad_enc += 1
print("encoder outputs that differ under concurrency:", bad_enc, "of", reps * 5, flush=True)

# 2. beam search of fixed tables while the other stream runs (a) the encoder (b) large GEMMs
for beam in beams:
    ref = []
    for q in quiet:
        t, n, s, _ = beam_search_ids(q, beam, cutoff, topn, 0, nbest=1, ext_scorer=scorer)
        torch.cuda.synchronize()
        ref.append((t.clone(), n.clone(), s.clone()))
    for load in os.environ.get("LOADS", "none,encoder,gemm").split(","):
        bad = 0
        for r in range(reps):
            for i, q in enumerate(quiet):
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    if load == "encoder":
                        x, la = batches[(i + 1) % 5]
                        model.get_encoder_out(x, la)
                    elif load == "gemm":
                        for _ in range(3):
                            big @ big
                t, n, s, _ = beam_search_ids(q, beam, cutoff, topn, 0, nbest=1, ext_scorer=scorer)
                torch.cuda.synchronize()
                if not (torch.equal(t, ref[i][0]) and torch.equal(n, ref[i][1]) and torch.equal(s, ref[i][2])):
                    bad += 1
        print(f"beam {beam} load {load}: decodes that differ: {bad} of {reps * 5}", flush=True)
