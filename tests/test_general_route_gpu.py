"""GPU: the general Conformer layer route (csrc/capi_generic.hip) on seeded random combinations of the ConformerEncoder
constructor arguments (conformer/encoder.py:38-48), ragged / degenerate batches and chunked streaming, against
oracle/conformer_oracle.py -- whose option branches are pinned to the reference's own source on the opt_* / act_* cases of
tests/golden/ref_small.npz (tests/test_ref_pin_cpu.py).  Tolerance as everywhere: 1e-3 of the tensor's largest magnitude
(north_star); measured values are printed."""
import numpy as np
import pytest
import torch

from oracle.conformer_oracle import ConformerOracle
from ppasr_amd.utils.synth import conformer_state_dict, synth_features

pytestmark = pytest.mark.gpu
TOL = 1e-3
# (hardshrink is left to its fixture, tests/test_ref_pin_gpu.py[act_hardshrink]: a DISCONTINUOUS activation -- x -> 0 below
#  |x| = 0.5 -- turns a 1e-7 difference in a pre-activation next to the threshold into a 0.5 step, so two correct fp32
#  implementations agree only as far as no value happens to sit there; seen here: 8e-3 on one draw)
ACTS = ["swish", "relu", "gelu", "tanh", "hardtanh", "relu6", "leakyrelu", "selu", "elu", "hardswish"]


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _draw(seed, streaming=False):
    """One random configuration: constructor arguments + model size (streaming: a causal model behind a conv front end)."""
    rng = np.random.Generator(np.random.PCG64(9000 + seed))
    causal = bool(rng.integers(2)) or streaming
    ks = int(rng.choice([3, 5, 9, 15, 31]) if not causal else rng.choice([2, 4, 8, 15, 16, 31]))
    width = int(rng.choice([256, 256, 512]))
    opts = dict(pos_enc_layer_type=str(rng.choice(["rel_pos", "abs_pos", "no_pos"])),
                normalize_before=bool(rng.integers(2)), concat_after=bool(rng.integers(2)),
                macaron_style=bool(rng.integers(2)), use_cnn_module=bool(rng.integers(4) > 0),
                activation_type=str(rng.choice(ACTS)))
    input_layer = str(rng.choice(["conv2d", "conv2d", "conv2d6", "conv2d8", "conv2d" if streaming else "linear"]))
    norm = str(rng.choice(["layer_norm", "batch_norm"]))
    return causal, ks, width, opts, input_layer, norm, rng


def _build(seed, streaming=False):
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    causal, ks, width, opts, input_layer, norm, rng = _draw(seed, streaming)
    V, L, heads = 53, int(rng.integers(1, 3)), width // 64
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=7000 + seed, perturb_norm=True, output_size=width,
                              attention_heads=heads, cnn_module_kernel=ks, cnn_module_norm=norm, input_layer=input_layer,
                              pos_enc_layer_type=opts["pos_enc_layer_type"], macaron_style=opts["macaron_style"],
                              use_cnn_module=opts["use_cnn_module"], concat_after=opts["concat_after"])
    conf = dict(output_size=width, attention_heads=heads, linear_units=2048, num_blocks=L, cnn_module_kernel=ks,
                cnn_module_norm=norm, input_layer=input_layer, **opts)
    model = ConformerModel(80, V, streaming=causal, encoder_conf=conf, state_dict=sd, device="cuda:0")
    oracle = ConformerOracle(sd, num_blocks=L, causal=causal, attention_heads=heads, cnn_module_kernel=ks, **opts)
    return model, oracle, conf, causal, input_layer, rng


@pytest.mark.parametrize("seed", range(14))
def test_random_options_batched(seed):
    model, oracle, conf, causal, input_layer, rng = _build(seed)
    t_min = {"linear": 1, "conv2d": 7, "conv2d6": 11, "conv2d8": 15}[input_layer]
    T = int(rng.integers(40, 90)) if input_layer == "linear" else int(rng.integers(60, 260))
    # a ragged batch with the degenerate lengths: full, random, 1 and 0 frames; and the shortest legal utterance alone
    for B, Tb, lens in ((4, T, [T, int(rng.integers(2, T)), 1, 0]), (1, t_min, [t_min])):
        x, la = synth_features(B, Tb, lens=lens, seed=seed + Tb)
        probs, logits = model.get_encoder_out(x, la, return_logits=True)
        ref_probs, ref_logits = oracle.get_encoder_out(x, la, return_logits=True)
        torch.cuda.synchronize()
        assert torch.isfinite(probs).all()
        e = _rel(logits.cpu().numpy(), ref_logits.numpy())
        print(f"seed {seed} {conf} B={B} T={Tb}: logits {e:.2e}")
        assert e < TOL, conf
        assert _rel(probs.cpu().numpy(), ref_probs.numpy()) < TOL


@pytest.mark.parametrize("seed", range(14))
def test_random_options_chunked(seed):
    model, oracle, conf, causal, input_layer, rng = _build(100 + seed, streaming=True)
    rate = {"conv2d": 4, "conv2d6": 6, "conv2d8": 8}[input_layer]
    required = int(rng.choice([-1, 0, 7, 20]))
    n = 4
    x, _ = synth_features(1, 600, seed=seed + 17)
    att = cnn = r_att = r_cnn = None
    off, pos = 0, 0
    for i in range(n):
        frames = int(rng.integers(16, 100)) if i else 67  # ragged chunk sizes (>= the front end's receptive field)
        frames = max(frames, 2 * rate + 7)
        p, att, cnn = model.get_encoder_out_chunk(x[:, pos:pos + frames], off, required, att, cnn)
        rp, r_att, r_cnn = oracle.get_encoder_out_chunk(x[:, pos:pos + frames], off, required, r_att, r_cnn)
        pos += frames
        off += p.shape[1]
        e = _rel(p.cpu().numpy(), rp.numpy())
        print(f"seed {seed} chunk {i} ({frames} frames, required {required}): probs {e:.2e}")
        assert e < TOL, conf
        assert tuple(att.shape) == tuple(r_att.shape), (att.shape, r_att.shape)
        if r_att.numel():
            assert _rel(att.cpu().numpy(), r_att.numpy()) < TOL
        assert tuple(cnn.shape) == tuple(r_cnn.shape), (cnn.shape, r_cnn.shape)
        if r_cnn.numel():
            assert _rel(cnn.cpu().numpy(), r_cnn.numpy()) < TOL


# ---- the other *former families at widths the fused kernels do not cover (Squeezeformer encoder_dim, Efficient-Conformer
# output_size 512: "for big data", configs/squeezeformer.yml:3-5, configs/efficient_conformer.yml:3-4) ----------------------
def _build_family(family, seed, streaming):
    from oracle.efficient_conformer_oracle import EfficientConformerOracle
    from oracle.squeezeformer_oracle import SqueezeformerOracle
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    from ppasr_amd.utils.synth import efficient_conformer_state_dict, squeezeformer_state_dict
    rng = np.random.Generator(np.random.PCG64(11000 + seed))
    V, L, width = 47, int(rng.integers(3, 5)), 512
    norm = str(rng.choice(["layer_norm", "batch_norm"]))
    if family == "squeezeformer":
        red = int(rng.integers(1, L - 1))
        rec = int(rng.integers(red + 1, L))
        ks = int(rng.choice([15, 31]))
        sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=12000 + seed, perturb_norm=True, streaming=streaming,
                                      cnn_norm_type=norm, encoder_dim=width, attention_heads=8, cnn_module_kernel=ks)
        conf = dict(encoder_dim=width, output_size=width, attention_heads=8, num_blocks=L, reduce_idx=red, recover_idx=rec,
                    feed_forward_expansion_factor=8, cnn_module_kernel=ks, cnn_norm_type=norm)
        model = SqueezeformerModel(80, V, streaming=streaming, encoder_conf=conf, state_dict=sd, device="cuda:0")
        oracle = SqueezeformerOracle(sd, num_blocks=L, reduce_idx=red, recover_idx=rec, causal=streaming, attention_heads=8,
                                     cnn_module_kernel=ks)
    else:
        stride_idx = int(rng.integers(0, L - 1))
        groups = tuple(sorted(set(int(v) for v in rng.integers(0, stride_idx + 1, size=2))))
        sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=12000 + seed, perturb_norm=True,
                                            stride_layer_idx=stride_idx, group_layer_idx=groups, output_size=width,
                                            attention_heads=8, cnn_module_norm=norm)
        conf = dict(output_size=width, attention_heads=8, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                    cnn_module_norm=norm,
                    efficient_conf=dict(stride_layer_idx=[stride_idx], stride=[2], group_layer_idx=list(groups), group_size=3,
                                        stride_kernel=True))
        model = EfficientConformerModel(80, V, streaming=streaming, encoder_conf=conf, state_dict=sd, device="cuda:0")
        oracle = EfficientConformerOracle(sd, num_blocks=L, stride_layer_idx=stride_idx, group_layer_idx=groups, causal=streaming,
                                          attention_heads=8)
    return model, oracle, conf, rng


@pytest.mark.parametrize("family", ["squeezeformer", "efficient_conformer"])
@pytest.mark.parametrize("seed", range(4))
def test_width_512_families_batched(family, seed):
    streaming = bool(seed & 1)
    model, oracle, conf, rng = _build_family(family, seed, streaming)
    T = int(rng.integers(70, 300))
    for B, Tb, lens in ((4, T, [T, int(rng.integers(2, T)), 1, 0]), (1, 7, [7])):
        x, la = synth_features(B, Tb, lens=lens, seed=seed + Tb)
        probs, logits = model.get_encoder_out(x, la, return_logits=True)
        ref_probs, ref_logits = oracle.get_encoder_out(x, la, return_logits=True)
        torch.cuda.synchronize()
        assert torch.isfinite(probs).all()
        e = _rel(logits.cpu().numpy(), ref_logits.numpy())
        print(f"{family} seed {seed} {conf} B={B} T={Tb}: logits {e:.2e}")
        assert e < TOL, conf


@pytest.mark.parametrize("family", ["squeezeformer", "efficient_conformer"])
@pytest.mark.parametrize("seed", range(3))
def test_width_512_families_chunked(family, seed):
    model, oracle, conf, rng = _build_family(family, 50 + seed, True)
    required = int(rng.choice([-16, 32, 16]))  # (even cache lengths: the reference's half-rate arithmetic needs them)
    # (whole 67-frame windows only: a short last window leaves an odd cache length, on which the Efficient-Conformer's
    #  export -- repeat_interleave of the half-rate caches -- fails in the reference and is refused here)
    x, _ = synth_features(1, 64 * 4 + 67, seed=seed + 23)
    att = cnn = r_att = r_cnn = None
    off = 0
    for cur in range(0, x.shape[1] - 7 + 1, 64):  # the predictor's 67-frame windows, stride 64 (predict.py:277-298)
        chunk = x[:, cur:min(cur + 67, x.shape[1])]
        p, att, cnn = model.get_encoder_out_chunk(chunk, off, required, att, cnn)
        rp, r_att, r_cnn = oracle.get_encoder_out_chunk(chunk, off, required, r_att, r_cnn)
        off += p.shape[1]
        e = _rel(p.cpu().numpy(), rp.numpy())
        print(f"{family} seed {seed} window {cur} required {required}: probs {e:.2e}")
        assert e < TOL, conf
        assert tuple(att.shape) == tuple(r_att.shape), (att.shape, r_att.shape)
        assert _rel(att.cpu().numpy(), r_att.numpy()) < TOL and _rel(cnn.cpu().numpy(), r_cnn.numpy()) < TOL


def test_ragged_decode_on_a_route_without_the_ragged_mode():
    """decode_ragged / evaluate(trim_padding) on a model whose route has no ragged mode (input_layer = linear: the library
    refuses skip_padding there, PPASR_EUNSUPPORTED; round 4 built it for the conv front ends of the general route): the
    drivers then run the padded rows and trim by frame_lens -- same tokens as the plain padded batch decoded over its valid
    frames."""
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    from ppasr_amd.parallel import decode_ragged, greedy_ids_decoder, set_skip_padding_if_built
    V, L = 53, 2
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=31, perturb_norm=True, output_size=512, attention_heads=8,
                              input_layer="linear")
    conf = dict(output_size=512, attention_heads=8, linear_units=2048, num_blocks=L, cnn_module_kernel=15, input_layer="linear")
    model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    assert set_skip_padding_if_built(model, True) is False
    wide = ConformerModel(80, V, streaming=True, encoder_conf=dict(conf, input_layer="conv2d"), device="cuda:0",
                          state_dict=conformer_state_dict(vocab_size=V, num_blocks=L, seed=31, perturb_norm=True, output_size=512,
                                                          attention_heads=8))
    assert set_skip_padding_if_built(wide, True) is True   # (the general route behind a conv front end has the mode now)
    wide.set_skip_padding(False)
    lens = np.array([260, 90, 411, 33, 187], np.int64)
    x, _ = synth_features(len(lens), int(lens.max()), lens=lens, seed=32)
    tokens, n, _ = decode_ragged(model, x, lens, greedy_ids_decoder(), mode="merged")
    torch.cuda.synchronize()
    # the same batch composition without the driver (an utterance's last frames depend on what it is padded with, so the
    # single-utterance call is not the reference here): padded batch -> valid frames -> greedy
    probs = model.get_encoder_out(x, lens)
    t1, n1, _ = greedy_ids_decoder()(probs, model.valid_out_frames(lens, x.shape[1]))
    for i in range(len(lens)):
        assert int(n[i]) > 0 and tokens[i, :int(n[i])].cpu().tolist() == t1[i, :int(n1[i])].cpu().tolist(), i
