"""cfg4's batch in both GEMM modes: logits, per-frame argmax, beam-search hypotheses and scores."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
from ppasr_amd.utils.synth import efficient_conformer_state_dict, synth_features
V = 4233
sd = efficient_conformer_state_dict(vocab_size=V, seed=1234)
conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12, cnn_module_kernel=15, cnn_module_norm="layer_norm",
            efficient_conf=dict(stride_layer_idx=[3], stride=[2], group_layer_idx=[0, 1, 2, 3], group_size=3))
m = EfficientConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
x, la = synth_features(64, 1000, seed=20240 + 400)
out = {}
for mode in ("f32", "f16x3"):
    m.set_gemm_mode(mode)
    probs, logits = m.get_encoder_out(x, la, return_logits=True)
    tok, n, score, _ = beam_search_ids(probs, 10, 0.99, 40, 0)
    torch.cuda.synchronize()
    out[mode] = (logits.cpu().numpy(), probs.cpu().numpy(), tok[:, 0].cpu().numpy(), n[:, 0].cpu().numpy(), score[:, 0].cpu().numpy())
a, b = out["f32"], out["f16x3"]
print("logits rel diff", float(np.abs(a[0] - b[0]).max() / np.abs(a[0]).max()), "probs max abs diff", float(np.abs(a[1] - b[1]).max()))
print("frames with a different argmax", int((a[0].argmax(-1) != b[0].argmax(-1)).sum()), "of", a[0].shape[0] * a[0].shape[1])
same = [bool(a[3][i] == b[3][i] and np.array_equal(a[2][i, :a[3][i]], b[2][i, :b[3][i]])) for i in range(64)]
print("utterances with the same best hypothesis", sum(same), "of 64; max |score diff|", float(np.abs(a[4] - b[4]).max()),
      "score magnitude", float(np.abs(a[4]).mean()))
for i in range(64):
    if not same[i]:
        d = [j for j in range(min(a[3][i], b[3][i])) if a[2][i, j] != b[2][i, j]]
        print("  utt", i, "lengths", a[3][i], b[3][i], "first differing position", d[:1], "scores", a[4][i], b[4][i])
