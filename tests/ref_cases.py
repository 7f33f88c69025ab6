"""Seeded cases shared by tests/golden/make_ref_goldens.py (which runs the REFERENCE's own model source on them through
oracle/paddle_shim) and by the tests that compare the oracles (CPU) and the HIP path (GPU) with the stored outputs.

Only outputs are stored in tests/golden/ref_*.npz; weights and features are regenerated from the seeds here."""
import numpy as np

from ppasr_amd.utils.synth import (conformer_state_dict, deepspeech2_state_dict, efficient_conformer_state_dict,
                                   squeezeformer_state_dict, synth_features)

WINDOW, STRIDE, CONTEXT = 67, 64, 7  # predict.py:277-283


def windows(n_frames, window=WINDOW, stride=STRIDE):
    """Chunk plan of PPASRPredictor.predict_stream with is_end=True (predict.py:291-298)."""
    return [(cur, min(cur + window, n_frames)) for cur in range(0, n_frames - CONTEXT + 1, stride)]


def _former(family, streaming, L, V, sd_seed, x_spec, chunk_frames=None, required=(), **kw):
    return dict(family=family, streaming=streaming, L=L, V=V, sd_seed=sd_seed, x_spec=x_spec, chunk_frames=chunk_frames,
                required=tuple(required), kw=kw)


# ---- small cases: every family, both `streaming` variants, ragged batches, chunked streaming -------------------------
SMALL = {
    # Conformer (configs/conformer.yml shape, fewer blocks)
    "conf_s": _former("conformer", True, 3, 61, 501, (3, 163, [163, 101, 37], 502), chunk_frames=64 * 4 + 30,
                      required=(-16, 32, 0)),
    "conf_n": _former("conformer", False, 2, 61, 503, (2, 131, [131, 77], 504)),
    # Efficient-Conformer: grouped attention on layers 0-1, stride layer 1, 7-tap kernels after it
    "eff_s": _former("efficient_conformer", True, 4, 53, 511, (2, 147, [147, 90], 512), chunk_frames=64 * 3 + 67,
                     required=(-16, 32), stride_layer_idx=1, group_layer_idx=(0, 1)),
    "eff_n": _former("efficient_conformer", False, 4, 53, 513, (2, 147, [147, 86], 514), stride_layer_idx=1,
                     group_layer_idx=(0, 1)),
    # output_size 512 / 8 heads (configs/efficient_conformer.yml:3-4): the general layer route (batched), an 11-tap conv
    "eff512_s": _former("efficient_conformer", True, 3, 53, 585, (3, 149, [149, 101, 9], 586), chunk_frames=64 * 3 + 67,
                        required=(-16, 32), stride_layer_idx=1, group_layer_idx=(0, 1), output_size=512, attention_heads=8),
    "eff512_n": _former("efficient_conformer", False, 3, 53, 587, (2, 148, [148, 77], 588), stride_layer_idx=1,
                        group_layer_idx=(0,), output_size=512, attention_heads=8, cnn_module_kernel=11),
    # output_size 768 / 12 heads: grouped attention on a width that is not a power of two (k_attention_t<192>'s flat-offset split)
    "eff768_n": _former("efficient_conformer", False, 2, 53, 589, (2, 148, [148, 77], 590), stride_layer_idx=1,
                        group_layer_idx=(0,), output_size=768, attention_heads=12),
    # the EfficientConformerEncoder constructor arguments no shipped YAML sets (efficient_conformer/encoder.py:50-54,117-128):
    # SEVERAL stride layers (kernels 15 -> 7 -> 3, 16x frame rate; the general layer route, batched) and group_size 2 / 4
    "eff_ms": _former("efficient_conformer", False, 5, 53, 591, (2, 197, [197, 120], 592), stride_layer_idx=[1, 3],
                      group_layer_idx=(0, 1, 2)),
    "eff_ms_c": _former("efficient_conformer", True, 4, 53, 593, (2, 163, [163, 77], 594), stride_layer_idx=[0, 2],
                        group_layer_idx=(0,)),
    "eff_g2": _former("efficient_conformer", False, 3, 53, 595, (2, 147, [147, 86], 596), stride_layer_idx=1,
                      group_layer_idx=(0, 1), group_size=2),
    "eff_g4": _former("efficient_conformer", True, 3, 53, 597, (3, 150, [150, 101, 13], 598), chunk_frames=64 * 3 + 67,
                      required=(-16, 32), stride_layer_idx=1, group_layer_idx=(0, 2), group_size=4),
    "eff512_g2": _former("efficient_conformer", False, 2, 53, 599, (2, 148, [148, 77], 600), stride_layer_idx=1,
                         group_layer_idx=(0, 1), group_size=2, output_size=512, attention_heads=8),
    # Squeezeformer: reduce before layer 1, recover before layer 3
    "sq_s": _former("squeezeformer", True, 4, 59, 521, (2, 131, [131, 77], 522), chunk_frames=64 * 4 + 40,
                    required=(-16, 32), reduce_idx=1, recover_idx=3),
    "sq_n": _former("squeezeformer", False, 4, 59, 523, (2, 131, [131, 70], 524), reduce_idx=1, recover_idx=3),
    # the SqueezeformerEncoder constructor arguments no shipped YAML sets (squeezeformer/encoder.py:28-44): adaptive_scale =
    # False with a depthwise second front-end conv (dw_stride), and final_proj (output_size != encoder_dim)
    "sq_opt": _former("squeezeformer", True, 3, 59, 525, (2, 131, [131, 77], 526), chunk_frames=64 * 3 + 40,
                      required=(-16,), reduce_idx=1, recover_idx=2, adaptive_scale=False, dw_stride=True),
    "sq_fproj": _former("squeezeformer", False, 3, 59, 527, (2, 131, [131, 70], 528), reduce_idx=1, recover_idx=2,
                        output_size=144),
    # conv-module BatchNorm1D instead of LayerNorm (cnn_module_norm / cnn_norm_type: batch_norm)
    "conf_bn": _former("conformer", True, 2, 61, 541, (2, 131, [131, 77], 542), chunk_frames=64 * 2 + 30,
                       required=(-16, 32), cnn_module_norm="batch_norm"),
    # 6x / 8x front ends (input_layer: conv2d6 / conv2d8): masks by 6t / 8t < len; chunks cut like the predictor does
    "conf6": _former("conformer", True, 2, 61, 545, (3, 197, [197, 120, 61], 546), chunk_frames=64 * 3 + 40,
                     required=(-16, 32), input_layer="conv2d6"),
    "conf8": _former("conformer", True, 2, 61, 547, (3, 203, [203, 150, 47], 548), chunk_frames=64 * 3 + 50,
                     required=(-16,), input_layer="conv2d8"),
    # a width the fused 256-column kernels do not cover: output_size 512 with 8 heads (the generic-width route)
    "conf512": _former("conformer", True, 2, 61, 549, (2, 131, [131, 77], 550), chunk_frames=64 * 3 + 30,
                       required=(-16, 32), output_size=512, attention_heads=8),
    "conf768_8": _former("conformer", False, 1, 61, 551, (2, 147, [147, 80], 552), output_size=768, attention_heads=12,
                         input_layer="conv2d8"),
    # ---- the ConformerEncoder constructor arguments no shipped YAML sets (conformer/encoder.py:38-48): the general
    # layer route.  Between them the cases cover every pos_enc_layer_type, normalize_before / concat_after /
    # macaron_style / use_cnn_module = both values, input_layer = linear, an unusual conv kernel, every activation ----
    "opt_abs": _former("conformer", True, 2, 61, 561, (2, 131, [131, 77], 562), chunk_frames=64 * 3 + 30, required=(-16, 32),
                       pos_enc_layer_type="abs_pos", activation_type="leakyrelu"),
    "opt_nopos_post": _former("conformer", True, 2, 61, 563, (2, 131, [131, 90], 564), chunk_frames=64 * 2 + 30,
                              required=(-16,), pos_enc_layer_type="no_pos", normalize_before=False, activation_type="relu"),
    "opt_nomac_concat": _former("conformer", False, 2, 61, 565, (3, 131, [131, 100, 1], 566), macaron_style=False,
                                concat_after=True, activation_type="gelu"),
    "opt_nocnn": _former("conformer", True, 2, 61, 567, (2, 131, [131, 77], 568), chunk_frames=64 * 2 + 30, required=(-16, 32),
                         use_cnn_module=False, activation_type="tanh"),
    "opt_linear": _former("conformer", False, 2, 61, 569, (2, 61, [61, 38], 570), input_layer="linear",
                          activation_type="hardswish"),
    "opt_k9": _former("conformer", False, 2, 61, 571, (2, 131, [131, 70], 572), cnn_module_kernel=9, activation_type="selu",
                      cnn_module_norm="batch_norm"),
    "opt_k5_post": _former("conformer", True, 1, 61, 573, (2, 99, [99, 50], 574), chunk_frames=64 * 2 + 30, required=(32,),
                           cnn_module_kernel=5, normalize_before=False, concat_after=True, pos_enc_layer_type="abs_pos",
                           activation_type="elu"),
    "act_hardtanh": _former("conformer", True, 1, 61, 575, (1, 99, [99], 576), activation_type="hardtanh"),
    # (31-tap causal conv: the 30 cached conv inputs outnumber a 16-frame chunk)
    "act_relu6": _former("conformer", True, 1, 61, 577, (1, 99, [99], 578), chunk_frames=64 * 2 + 30, required=(-16,),
                         activation_type="relu6", cnn_module_kernel=31),
    "act_hardshrink": _former("conformer", True, 1, 61, 579, (1, 99, [99], 580), activation_type="hardshrink"),
    # activation_type (squeezeformer/encoder.py:45) other than swish: the general layer route at width 256
    "sq_gelu_s": _former("squeezeformer", True, 3, 59, 601, (2, 131, [131, 77], 602), chunk_frames=64 * 3 + 40, required=(-16,),
                         reduce_idx=1, recover_idx=2, activation_type="gelu"),
    "sq_relu_n": _former("squeezeformer", False, 3, 59, 603, (2, 131, [131, 70], 604), reduce_idx=1, recover_idx=2,
                         activation_type="relu", cnn_norm_type="batch_norm"),
    # normalize_before = True (squeezeformer/encoder.py:49): pre-norm layers
    "sq_pre_s": _former("squeezeformer", True, 3, 59, 605, (2, 131, [131, 77], 606), chunk_frames=64 * 3 + 40, required=(-16,),
                        reduce_idx=1, recover_idx=2, normalize_before=True),
    "sq_pre_n": _former("squeezeformer", False, 3, 59, 607, (2, 131, [131, 70], 608), reduce_idx=1, recover_idx=2,
                        normalize_before=True, activation_type="hardswish"),
    # pos_enc_layer_type != rel_pos (squeezeformer/encoder.py:101-105): conformer's plain MultiHeadedAttention
    "sq_abs_s": _former("squeezeformer", True, 3, 59, 609, (2, 131, [131, 77], 610), chunk_frames=64 * 3 + 40, required=(-16, 32),
                        reduce_idx=1, recover_idx=2, pos_enc_layer_type="abs_pos"),
    "sq_nopos_n": _former("squeezeformer", False, 3, 59, 611, (2, 131, [131, 70], 612), reduce_idx=1, recover_idx=2,
                          pos_enc_layer_type="no_pos", normalize_before=True),
    "sq_bn": _former("squeezeformer", False, 3, 59, 543, (2, 131, [131, 70], 544), reduce_idx=1, recover_idx=2,
                     cnn_norm_type="batch_norm"),
    # encoder_dim 512 / 8 heads (configs/squeezeformer.yml:3-5 "for big data ... 512"): the general layer route
    "sq512_s": _former("squeezeformer", True, 3, 59, 581, (3, 131, [131, 90, 5], 582), chunk_frames=64 * 4 + 40,
                       required=(-16, 32), reduce_idx=1, recover_idx=2, encoder_dim=512, attention_heads=8),
    "sq512_n": _former("squeezeformer", False, 3, 59, 583, (2, 147, [147, 70], 584), reduce_idx=1, recover_idx=2,
                       encoder_dim=512, attention_heads=8, cnn_norm_type="batch_norm"),
}
for _gru in (False, True):
    for _streaming in (True, False):
        _k = "ds2" + ("g" if _gru else "") + ("_s" if _streaming else "_b")
        SMALL[_k] = dict(family="deepspeech2", streaming=_streaming, L=2, V=47, sd_seed=531 + 2 * _gru + _streaming,
                         x_spec=(3, 99, [99, 64, 31], 535), chunk_frames=(64 * 2 + 40) if _streaming else None,
                         required=(), kw=dict(use_gru=_gru))

# ---- BASELINE.json configs at full size (outputs stored sub-sampled, see make_ref_goldens.py) ------------------------
FULL = {
    # configs[0]: DeepSpeech2 non-streaming (bidirectional LSTM, deepspeech2/encoder.py:61-104), 5 x 1024, one 5 s utterance
    "cfg1": dict(family="deepspeech2", streaming=False, L=5, V=4233, sd_seed=1234, x_spec=(1, 498, None, 20240 + 100),
                 chunk_frames=None, required=(), kw=dict(use_gru=False)),
    # configs[1]: Conformer streaming, 32 x 1000 frames, V = 4233 (bench.py's workload, rank 0 seed)
    "cfg2": _former("conformer", True, 12, 4233, 1234, (32, 1000, None, 20240 + 200)),
    # configs[3]: Efficient-Conformer streaming, B = 64, beam 10 / 0.99 / top-40
    "cfg4": _former("efficient_conformer", True, 12, 4233, 1234, (64, 1000, None, 20240 + 400)),
    # configs[4]: Squeezeformer streaming, one GPU's share (16) of B = 128, lengths U{200..3000}
    "cfg5": _former("squeezeformer", True, 12, 4233, 1234, (16, None, "ragged", 20240 + 500)),
}
BEAM = dict(beam_size=10, cutoff_prob=0.99, cutoff_top_n=40)
SAMPLED_COLUMNS = 32


def cfg5_lengths(n=16, seed=20240 + 500):
    rng = np.random.Generator(np.random.PCG64(seed))
    return np.sort(rng.integers(200, 3001, size=n))[::-1].astype(np.int64)


def bucket_of(length, width=200):
    """200-frame length buckets, a bucket is padded to its longest member."""
    return (int(length) - 1) // width


def sampled_columns(V, n=SAMPLED_COLUMNS):
    return np.unique(np.concatenate([[0, 1, V - 1], np.linspace(2, V - 2, n - 3).astype(np.int64)]))


def state_dict(case, perturb=True):
    fam, L, V, seed, kw = case["family"], case["L"], case["V"], case["sd_seed"], case["kw"]
    full = L == 12
    pn = perturb and not full
    if fam == "conformer":
        return conformer_state_dict(vocab_size=V, num_blocks=L, seed=seed, perturb_norm=pn,
                                    **{k: v for k, v in kw.items() if k not in ("activation_type", "normalize_before")})
    if fam == "efficient_conformer":
        if full:
            return efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=seed)
        return efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=seed, perturb_norm=pn,
                                              stride_layer_idx=kw["stride_layer_idx"], group_layer_idx=kw["group_layer_idx"],
                                              group_size=kw.get("group_size", 3),
                                              output_size=kw.get("output_size", 256), attention_heads=kw.get("attention_heads", 4),
                                              cnn_module_kernel=kw.get("cnn_module_kernel", 15))
    if fam == "squeezeformer":
        return squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=seed, perturb_norm=pn, streaming=case["streaming"],
                                        cnn_norm_type=kw.get("cnn_norm_type", "layer_norm"),
                                        encoder_dim=kw.get("encoder_dim", 256), attention_heads=kw.get("attention_heads", 4),
                                        dw_stride=kw.get("dw_stride", False), output_size=kw.get("output_size"),
                                        plain_mha=kw.get("pos_enc_layer_type", "rel_pos") != "rel_pos")
    if fam == "deepspeech2":
        return deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, streaming=case["streaming"], seed=seed,
                                      perturb_norm=pn, use_gru=kw.get("use_gru", False))
    raise ValueError(fam)


def features(case):
    B, T, lens, seed = case["x_spec"]
    if lens == "ragged":
        lens = cfg5_lengths(B, seed)
        return synth_features(B, int(lens.max()), lens=lens, seed=seed)
    return synth_features(B, T, lens=lens, seed=seed)


def chunk_features(case):
    x, _ = synth_features(3 if case["family"] == "deepspeech2" else 1, case["chunk_frames"], seed=case["sd_seed"] + 50)
    return x


def reference_encoder_conf(case):
    """encoder_conf exactly as the reference's constructors take it (flat keyword arguments; note that the nested
    `efficient_conf:` block of configs/efficient_conformer.yml lands in **kwargs of EfficientConformerEncoder.__init__
    and is ignored there -- efficient_conformer/encoder.py:26-56 -- the shipped values equal the defaults)."""
    fam, L, kw = case["family"], case["L"], case["kw"]
    if fam == "conformer":
        c = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, dropout_rate=0.1,
                 positional_dropout_rate=0.1, attention_dropout_rate=0.1, input_layer="conv2d", normalize_before=True,
                 cnn_module_kernel=15, use_cnn_module=True, activation_type="swish", pos_enc_layer_type="rel_pos",
                 cnn_module_norm="layer_norm")
        c.update(kw)  # (concat_after / macaron_style appear only where a case sets them)
        return c
    if fam == "efficient_conformer":
        c = dict(output_size=kw.get("output_size", 256), attention_heads=kw.get("attention_heads", 4), linear_units=2048,
                 num_blocks=L, activation_type="swish", cnn_module_kernel=kw.get("cnn_module_kernel", 15),
                 cnn_module_norm="layer_norm", dropout_rate=0.1, input_layer="conv2d",
                 normalize_before=True, pos_enc_layer_type="rel_pos", attention_dropout_rate=0.1,
                 positional_dropout_rate=0.1)
        if L != 12:
            sl = kw["stride_layer_idx"]
            c.update(stride_layer_idx=sl, stride=2 if isinstance(sl, int) else [2] * len(sl),
                     group_layer_idx=tuple(kw["group_layer_idx"]), group_size=kw.get("group_size", 3), stride_kernel=True)
        else:
            c["efficient_conf"] = dict(stride_layer_idx=[3], stride=[2], group_layer_idx=[0, 1, 2, 3], group_size=3,
                                       stride_kernel=True)
        return c
    if fam == "squeezeformer":
        return dict(encoder_dim=kw.get("encoder_dim", 256), output_size=kw.get("output_size", kw.get("encoder_dim", 256)),
                    attention_heads=kw.get("attention_heads", 4), num_blocks=L,
                    reduce_idx=kw.get("reduce_idx", 5), recover_idx=kw.get("recover_idx", 11),
                    feed_forward_expansion_factor=8, input_dropout_rate=0.1, feed_forward_dropout_rate=0.1,
                    attention_dropout_rate=0.1, adaptive_scale=kw.get("adaptive_scale", True),
                    dw_stride=kw.get("dw_stride", False), cnn_module_kernel=31, normalize_before=kw.get("normalize_before", False),
                    activation_type=kw.get("activation_type", "swish"), pos_enc_layer_type=kw.get("pos_enc_layer_type", "rel_pos"),
                    cnn_norm_type=kw.get("cnn_norm_type", "layer_norm"))
    if fam == "deepspeech2":
        return dict(num_rnn_layers=L, rnn_size=1024, use_gru=kw.get("use_gru", False))
    raise ValueError(fam)


def product_encoder_conf(case):
    """The same configuration in the form ppasr_amd's model wrappers take (YAML-shaped)."""
    c = reference_encoder_conf(case)
    if case["family"] == "efficient_conformer" and "efficient_conf" not in c:
        sl, st = c.pop("stride_layer_idx"), c.pop("stride")
        c["efficient_conf"] = dict(stride_layer_idx=[sl] if isinstance(sl, int) else list(sl),
                                   stride=[st] if isinstance(st, int) else list(st),
                                   group_layer_idx=list(c.pop("group_layer_idx")), group_size=c.pop("group_size"),
                                   stride_kernel=c.pop("stride_kernel"))
    return c
