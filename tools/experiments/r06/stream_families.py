"""One 0.64 s chunk of one streaming session, the three *former families at their shipped depth (12 blocks): ms per chunk
(best of 5 rounds of 40 chunks)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from ppasr_amd.utils.synth import (DEFAULT_VOCAB_SIZE, conformer_state_dict, efficient_conformer_state_dict,
                                   squeezeformer_state_dict, synth_features)

V, L = DEFAULT_VOCAB_SIZE, 12


def build(fam):
    if fam == "conformer":
        from ppasr_amd.model_utils.conformer.model import ConformerModel
        sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=1)
        conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
        return ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    if fam == "efficient":
        from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
        sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=2, stride_layer_idx=3, group_layer_idx=(0, 1, 2, 3))
        conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                    cnn_module_norm="layer_norm",
                    efficient_conf=dict(stride_layer_idx=[3], stride=[2], group_layer_idx=[0, 1, 2, 3], group_size=3,
                                        stride_kernel=True))
        return EfficientConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=3)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=5, recover_idx=11,
                feed_forward_expansion_factor=8, cnn_module_kernel=31)
    return SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")


x, _ = synth_features(1, 67, seed=5)
chunk = torch.from_numpy(x).cuda()
out = {"tag": os.environ.get("TAG", "")}
for fam in os.environ.get("FAMS", "conformer,efficient,squeezeformer").split(","):
    model = build(fam)
    s = model.new_stream()
    best = 1e9
    for rep in range(6):
        s.reset()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(40):
            s.encode_chunk(chunk, 32, want_probs=False, want_frames=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 40 * 1e3
        if rep:
            best = min(best, dt)
    out[fam + "_ms"] = round(best, 3)
print(json.dumps(out), flush=True)
