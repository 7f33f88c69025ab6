"""Ad-hoc fuzz of forward_chunk: random chunk lengths / required_cache_size per call, all three *former families, GPU vs the
oracles' restatements of the reference (cases on which the reference's own shape arithmetic fails must be refused too)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.conformer_oracle import ConformerOracle
from oracle.efficient_conformer_oracle import EfficientConformerOracle
from oracle.squeezeformer_oracle import SqueezeformerOracle
from ppasr_amd.utils.synth import conformer_state_dict, efficient_conformer_state_dict, squeezeformer_state_dict, synth_features

V = 90
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

def build(family, seed):
    if family == "conformer":
        from ppasr_amd.model_utils.conformer.model import ConformerModel
        sd = conformer_state_dict(vocab_size=V, num_blocks=2, seed=seed, perturb_norm=True)
        conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=2, cnn_module_kernel=15)
        return ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd), ConformerOracle(sd, num_blocks=2)
    if family == "squeezeformer":
        from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
        sd = squeezeformer_state_dict(vocab_size=V, num_blocks=4, seed=seed, perturb_norm=True)
        conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=4, reduce_idx=1, recover_idx=3,
                    feed_forward_expansion_factor=8, cnn_module_kernel=31)
        return (SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd),
                SqueezeformerOracle(sd, num_blocks=4, reduce_idx=1, recover_idx=3))
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=4, seed=seed, perturb_norm=True, stride_layer_idx=1,
                                        group_layer_idx=(0, 1))
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=4, cnn_module_kernel=15,
                cnn_module_norm="layer_norm", efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1],
                                                                  group_size=3, stride_kernel=True))
    return (EfficientConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd),
            EfficientConformerOracle(sd, num_blocks=4, stride_layer_idx=1, group_layer_idx=(0, 1)))

bad = 0; n_ok = 0; n_refused = 0
for family in ("conformer", "squeezeformer", "efficient"):
    for seed in range(8):
        rng = np.random.Generator(np.random.PCG64(seed * 17 + len(family)))
        model, oracle = build(family, seed)
        stream = model.new_stream()
        att = cnn = None; offset = 0
        for step in range(6):
            T = int(rng.choice([67, 67, 67, 35, 51, 99, 131, 23, 8, 70]))
            req = int(rng.choice([-16, -16, -1, 0, 16, 32, 48, 7, 33]))
            x, _ = synth_features(1, T, seed=seed * 100 + step)
            ref_err = got_err = None
            try:
                ref, att, cnn = oracle.get_encoder_out_chunk(x, offset, req, att, cnn)
            except Exception as e:
                ref_err = e
            try:
                got = stream.encode_chunk(x, req)
                g_att, g_cnn = stream.export_caches()
                torch.cuda.synchronize()
            except Exception as e:
                got_err = e
            if ref_err is not None or got_err is not None:
                n_refused += 1
                if (ref_err is None) != (got_err is None):
                    bad += 1
                    print("DISAGREE", family, seed, step, T, req, "ref:", repr(ref_err)[:80], "got:", repr(got_err)[:80])
                break  # state undefined afterwards
            e1 = rel(got.cpu().numpy(), ref.numpy())
            e2 = rel(g_att.cpu().numpy(), att.numpy()) if att.shape[2] > 0 else 0.0
            ok = tuple(got.shape) == tuple(ref.shape) and e1 < 1e-3 and tuple(g_att.shape) == tuple(att.shape) and e2 < 1e-3
            if not ok:
                bad += 1
                print("MISMATCH", family, seed, step, T, req, got.shape, ref.shape, e1, g_att.shape, att.shape, e2)
                break
            n_ok += 1
            offset += ref.shape[1]
print("fuzz_stream done: ok chunks", n_ok, "refused by both", n_refused - bad, "problems", bad)
