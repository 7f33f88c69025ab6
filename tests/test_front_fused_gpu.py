"""GPU: Conv2dSubsampling4's two convolutions as one launch (csrc/front_fused.hip: conv1's output is computed tile by tile
inside conv2's implicit GEMM and never written) against the two-launch route it replaces (ppasr_set_front_fused(0): k_conv1 +
k_gemm_stream<conv2>) -- the same fmaf chain per conv1 element and the same MFMA order: bit-identical logits, for every
tile size of the launch logic (32 / 64 / 96 / 128-row tiles, whole rounds + a re-cut remainder, the active-tile table of
ragged batches)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(front_fused):
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    from ppasr_amd.utils.synth import conformer_state_dict, squeezeformer_state_dict, synth_features
    V = 97
    sd = conformer_state_dict(vocab_size=V, num_blocks=1, seed=91, perturb_norm=True)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=1, cnn_module_kernel=15)
    cm = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    sq_sd = squeezeformer_state_dict(vocab_size=V, num_blocks=2, seed=92, perturb_norm=True)
    sq_conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=2, reduce_idx=None, recover_idx=None,
                   feed_forward_expansion_factor=8, cnn_module_kernel=31)
    sm = SqueezeformerModel(80, V, streaming=True, encoder_conf=sq_conf, state_dict=sq_sd, device="cuda:0")
    cm.set_front_fused(front_fused)
    sm.set_front_fused(front_fused)
    out = {}
    # (B, T): conv2 rows B * T' * 19 -> 1 x 67: 304 rows (32-row tiles); 3 x 203: 2850 (32); 9 x 403: 17100 (96-row tiles);
    # 16 x 611: 46208 (128-row tiles + a 96-row remainder); 33 x 1000: 156123 (4 rounds + remainder)
    for B, T in ((1, 67), (3, 203), (9, 403), (16, 611), (33, 1000)):
        rng = np.random.default_rng(B)
        lens = [T] + [int(v) for v in rng.integers(40, T + 1, size=B - 1)]
        x, la = synth_features(B, T, lens=lens, seed=93 + B)
        for skip in (False, True):
            cm.set_skip_padding(skip)
            out["c_%d_%d_%d" % (B, T, skip)] = cm.get_encoder_out(x, la, return_logits=True)[1].cpu().numpy()
        cm.set_skip_padding(False)
    x, la = synth_features(6, 611, lens=[611, 600, 333, 97, 611, 13], seed=94)
    out["s"] = sm.get_encoder_out(x, la, return_logits=True)[1].cpu().numpy()
    return out


def test_one_launch_front_end_is_bit_identical_to_conv1_then_conv2():
    outs = [_run(1), _run(0)]
    assert sorted(outs[0]) == sorted(outs[1]) and len(outs[0]) == 11
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))
