"""Seeded synthetic checkpoints / features / vocabularies for the hot path.

No PPASR checkpoint is reachable offline, so tests and bench.py
run random-init weights drawn with the reference's own initialisers:

* ``ppasr/model_utils/utils/base.py:58-78`` (``Linear``/``Conv1D``/``Conv2D``:
  KaimingUniform(negative_slope=sqrt(5)) -> U(+-1/sqrt(fan_in)))
* plain ``nn.Linear`` for ``embed.out`` (``conformer/subsampling.py:87``) and
  ``ctc_lo`` (``loss/ctc.py:27``): Xavier-uniform weight, zero bias
* ``pos_bias_u/v``: Xavier-uniform (``conformer/attention.py:193-196``)
* LayerNorm gamma=1, beta=0 (``base.py:7-21``)

Parameter names and layouts are the Paddle ones (``Linear.weight`` is
``[in, out]``), i.e. exactly what a ``model.pdparams`` state dict holds, so a
real checkpoint can be dropped in later.  Everything is numpy float32.
"""
import math

import numpy as np

__all__ = ["conformer_state_dict", "squeezeformer_state_dict", "efficient_conformer_state_dict", "deepspeech2_state_dict", "synth_features", "synth_vocabulary", "DEFAULT_VOCAB_SIZE"]

DEFAULT_VOCAB_SIZE = 4233  # <blank>, <unk>, 4230 CJK chars, <eos>  


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _kaiming(rng, shape, fan_in):
    return _uniform(rng, shape, 1.0 / math.sqrt(fan_in))


def _xavier(rng, shape, fan_in, fan_out):
    return _uniform(rng, shape, math.sqrt(6.0 / (fan_in + fan_out)))


def _layernorm(sd, prefix, size, rng, perturb):
    if perturb:
        sd[prefix + ".weight"] = (1.0 + 0.1 * rng.standard_normal(size)).astype(np.float32)
        sd[prefix + ".bias"] = (0.1 * rng.standard_normal(size)).astype(np.float32)
    else:
        sd[prefix + ".weight"] = np.ones(size, np.float32)
        sd[prefix + ".bias"] = np.zeros(size, np.float32)


def _batchnorm(sd, prefix, size, rng):
    """nn.BatchNorm1D inference parameters (cnn_module_norm: batch_norm): affine + running statistics, all drawn away
    from the (1, 0, 0, 1) initial values so that a dropped term shows."""
    sd[prefix + ".weight"] = (1.0 + 0.1 * rng.standard_normal(size)).astype(np.float32)
    sd[prefix + ".bias"] = (0.1 * rng.standard_normal(size)).astype(np.float32)
    sd[prefix + "._mean"] = (0.3 * rng.standard_normal(size)).astype(np.float32)
    sd[prefix + "._variance"] = rng.uniform(0.5, 1.5, size).astype(np.float32)


def _linear(sd, prefix, fin, fout, rng, bias=True):
    # base.Linear: Paddle layout [in, out]; fan_in = shape[0]
    sd[prefix + ".weight"] = _kaiming(rng, (fin, fout), fin)
    if bias:
        # KaimingUniform on a 1-D tensor: fan_in = shape[0]
        sd[prefix + ".bias"] = _kaiming(rng, (fout,), fout)


def conformer_state_dict(input_dim=80, vocab_size=DEFAULT_VOCAB_SIZE, output_size=256, attention_heads=4,
                         linear_units=2048, num_blocks=12, cnn_module_kernel=15, seed=1234,
                         ctc_sharpen=8.0, perturb_norm=False, cmvn_mean=10.0, cmvn_istd=1.0 / 3.3,
                         cnn_module_norm="layer_norm", input_layer="conv2d", pos_enc_layer_type="rel_pos",
                         macaron_style=True, use_cnn_module=True, concat_after=False):
    """Random-init ``ConformerModel`` inference parameters (encoder + CTC head).

    ``ctc_sharpen`` multiplies ``ctc.ctc_lo.weight`` so that greedy top-1 margins
    are realistic for un-trained weights (documented deviation).
    ``perturb_norm`` draws LayerNorm gamma/beta away from (1, 0) so tests notice
    a dropped affine term.  The keyword arguments behind ``input_layer`` follow ``ConformerEncoder.__init__``
    (conformer/encoder.py:38-48): the parameter SET changes with them exactly as the reference's ``state_dict()`` does
    (checked by loading the dict by name into the reference classes, tests/golden/make_ref_goldens.py).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    d, h = output_size, attention_heads
    dk = d // h
    f1 = (input_dim - 1) // 2
    sd = {}
    sd["encoder.global_cmvn.mean"] = np.full(input_dim, cmvn_mean, np.float32)
    sd["encoder.global_cmvn.istd"] = np.full(input_dim, cmvn_istd, np.float32)
    if perturb_norm:
        sd["encoder.global_cmvn.mean"] += (0.5 * rng.standard_normal(input_dim)).astype(np.float32)
        sd["encoder.global_cmvn.istd"] *= (1.0 + 0.1 * rng.uniform(-1, 1, input_dim)).astype(np.float32)
    if input_layer == "linear":  # LinearNoSubsampling (subsampling.py:39-42): nn.Linear, nn.LayerNorm, Dropout, ReLU
        _linear(sd, "encoder.embed.out.0", input_dim, d, rng)
        _layernorm(sd, "encoder.embed.out.1", d, rng, perturb_norm)
    else:
        # Conv2dSubsampling4 / 6 / 8 (subsampling.py): base.Conv2D (Kaiming, fan_in = Cin*kh*kw)
        sd["encoder.embed.conv.0.weight"] = _kaiming(rng, (d, 1, 3, 3), 9)
        sd["encoder.embed.conv.0.bias"] = _kaiming(rng, (d,), d)
    if input_layer == "linear":
        pass
    elif input_layer == "conv2d":
        f_last, lin = (f1 - 1) // 2, "encoder.embed.out.0"
        sd["encoder.embed.conv.2.weight"] = _kaiming(rng, (d, d, 3, 3), d * 9)
        sd["encoder.embed.conv.2.bias"] = _kaiming(rng, (d,), d)
    elif input_layer == "conv2d6":  # Conv2D(odim, odim, 5, 3); projection named `linear`
        f_last, lin = (f1 - 2) // 3, "encoder.embed.linear"
        sd["encoder.embed.conv.2.weight"] = _kaiming(rng, (d, d, 5, 5), d * 25)
        sd["encoder.embed.conv.2.bias"] = _kaiming(rng, (d,), d)
    elif input_layer == "conv2d8":  # three 3x3 / 2 convs
        f_last, lin = (((f1 - 1) // 2) - 1) // 2, "encoder.embed.linear"
        for idx in (2, 4):
            sd[f"encoder.embed.conv.{idx}.weight"] = _kaiming(rng, (d, d, 3, 3), d * 9)
            sd[f"encoder.embed.conv.{idx}.bias"] = _kaiming(rng, (d,), d)
    else:
        raise ValueError(input_layer)
    if input_layer != "linear":
        sd[lin + ".weight"] = _xavier(rng, (d * f_last, d), d * f_last, d)
        sd[lin + ".bias"] = np.zeros(d, np.float32)
    for i in range(num_blocks):
        p = f"encoder.encoders.{i}."
        for name in ("linear_q", "linear_k", "linear_v", "linear_out"):
            _linear(sd, p + "self_attn." + name, d, d, rng)
        if pos_enc_layer_type == "rel_pos":  # RelPositionMultiHeadedAttention; abs_pos / no_pos: MultiHeadedAttention
            _linear(sd, p + "self_attn.linear_pos", d, d, rng, bias=False)
            sd[p + "self_attn.pos_bias_u"] = _xavier(rng, (h, dk), h, dk)
            sd[p + "self_attn.pos_bias_v"] = _xavier(rng, (h, dk), h, dk)
        for ff in ("feed_forward", "feed_forward_macaron") if macaron_style else ("feed_forward",):
            _linear(sd, p + ff + ".w_1", d, linear_units, rng)
            _linear(sd, p + ff + ".w_2", linear_units, d, rng)
        if use_cnn_module:
            # ConvolutionModule: base.Conv1D, weight [out, in/groups, k]
            sd[p + "conv_module.pointwise_conv1.weight"] = _kaiming(rng, (2 * d, d, 1), d)
            sd[p + "conv_module.pointwise_conv1.bias"] = _kaiming(rng, (2 * d,), 2 * d)
            sd[p + "conv_module.depthwise_conv.weight"] = _kaiming(rng, (d, 1, cnn_module_kernel), cnn_module_kernel)
            sd[p + "conv_module.depthwise_conv.bias"] = _kaiming(rng, (d,), d)
            if cnn_module_norm == "batch_norm":
                _batchnorm(sd, p + "conv_module.norm", d, rng)
            else:
                _layernorm(sd, p + "conv_module.norm", d, rng, perturb_norm)
            sd[p + "conv_module.pointwise_conv2.weight"] = _kaiming(rng, (d, d, 1), d)
            sd[p + "conv_module.pointwise_conv2.bias"] = _kaiming(rng, (d,), d)
        norms = ["norm_ff", "norm_mha"] + (["norm_ff_macaron"] if macaron_style else []) + \
                (["norm_conv", "norm_final"] if use_cnn_module else [])
        for n in norms:
            _layernorm(sd, p + n, d, rng, perturb_norm)
        if concat_after:  # encoder.py:341-342
            _linear(sd, p + "concat_linear", 2 * d, d, rng)
    _layernorm(sd, "encoder.after_norm", d, rng, perturb_norm)
    sd["ctc.ctc_lo.weight"] = _xavier(rng, (d, vocab_size), d, vocab_size) * np.float32(ctc_sharpen)
    sd["ctc.ctc_lo.bias"] = np.zeros(vocab_size, np.float32)
    if perturb_norm:
        sd["ctc.ctc_lo.bias"] = (0.1 * rng.standard_normal(vocab_size)).astype(np.float32)
    return sd


def synth_features(batch, frames, n_mels=80, lens=None, seed=20240, mean=10.0, std=3.3):
    """``feats = mean + std*N(0,1)`` float32 ``[B, T, F]``; rows >= len are zero
    (the reference's collate_fn zero-pads, ``data_utils/collate_fn.py:17``)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = (mean + std * rng.standard_normal((batch, frames, n_mels))).astype(np.float32)
    if lens is None:
        lens = np.full(batch, frames, np.int64)
    lens = np.asarray(lens, np.int64)
    for b in range(batch):
        x[b, int(lens[b]):] = 0.0
    return x, lens


def synth_vocabulary(vocab_size=DEFAULT_VOCAB_SIZE):
    """``<blank>`` (id 0), ``<unk>``, CJK characters, ``<eos>`` last
    (vocabulary layout of ``ppasr/trainer.py:480-487``)."""
    chars = [chr(0x4E00 + i) for i in range(vocab_size - 3)]
    return ["<blank>", "<unk>"] + chars + ["<eos>"]


def squeezeformer_state_dict(input_dim=80, vocab_size=DEFAULT_VOCAB_SIZE, encoder_dim=256, attention_heads=4,
                             feed_forward_expansion_factor=8, num_blocks=12, cnn_module_kernel=31, seed=1234,
                             ctc_sharpen=8.0, perturb_norm=False, cmvn_mean=10.0, cmvn_istd=1.0 / 3.3, streaming=True,
                             cnn_norm_type="layer_norm", dw_stride=False, output_size=None, plain_mha=False):
    """Random-init ``SqueezeformerModel`` inference parameters (``streaming=False``: the time-reduction layer is
    ``TimeReductionLayer1D`` with a 5-tap depthwise conv instead of the 1-tap ``TimeReductionLayerStream``,
    squeezeformer/model.py:35-39).  The reference's ``init_weights()`` calls have
    no effect on the values: every layer keeps the ``utils/base.py`` Kaiming-uniform init;
    attention / conv ``ada_scale``/``ada_bias`` are 1/0, the FFN ones are Xavier-uniform
    (squeezeformer/positionwise.py:39-42).  ``perturb_norm`` randomises every norm / adaptive-scale vector.
    ``dw_stride``: the front end's second conv is depthwise (subsampling.py:38); ``output_size`` != ``encoder_dim``: the
    encoder ends in ``final_proj`` and the CTC head reads ``output_size`` features (encoder.py:165-167)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d, h = encoder_dim, attention_heads
    dk = d // h
    ff = d * feed_forward_expansion_factor
    f2 = ((input_dim - 1) // 2 - 1) // 2
    sd = {}
    sd["encoder.global_cmvn.mean"] = np.full(input_dim, cmvn_mean, np.float32)
    sd["encoder.global_cmvn.istd"] = np.full(input_dim, cmvn_istd, np.float32)
    sd["encoder.embed.pw_conv.weight"] = _kaiming(rng, (d, 1, 3, 3), 9)
    sd["encoder.embed.pw_conv.bias"] = _kaiming(rng, (d,), d)
    sd["encoder.embed.dw_conv.weight"] = _kaiming(rng, (d, 1, 3, 3), 9) if dw_stride else _kaiming(rng, (d, d, 3, 3), d * 9)
    sd["encoder.embed.dw_conv.bias"] = _kaiming(rng, (d,), d)
    _linear(sd, "encoder.embed.input_proj.0", d * f2, d, rng)
    _layernorm(sd, "encoder.preln", d, rng, perturb_norm)

    def ada(prefix, xavier):
        if xavier:
            b = math.sqrt(6.0 / (2 * d))
            sd[prefix + ".ada_scale"] = _uniform(rng, (1, 1, d), b)
            sd[prefix + ".ada_bias"] = _uniform(rng, (1, 1, d), b)
        elif perturb_norm:
            sd[prefix + ".ada_scale"] = (1.0 + 0.1 * rng.standard_normal((1, 1, d))).astype(np.float32)
            sd[prefix + ".ada_bias"] = (0.1 * rng.standard_normal((1, 1, d))).astype(np.float32)
        else:
            sd[prefix + ".ada_scale"] = np.ones((1, 1, d), np.float32)
            sd[prefix + ".ada_bias"] = np.zeros((1, 1, d), np.float32)

    for i in range(num_blocks):
        p = f"encoder.encoders.{i}."
        for name in ("linear_q", "linear_k", "linear_v", "linear_out") + (() if plain_mha else ("linear_pos",)):
            _linear(sd, p + "self_attn." + name, d, d, rng)
        if not plain_mha:  # (pos_enc_layer_type != rel_pos: conformer's MultiHeadedAttention has none of these)
            sd[p + "self_attn.pos_bias_u"] = _xavier(rng, (h, dk), h, dk)
            sd[p + "self_attn.pos_bias_v"] = _xavier(rng, (h, dk), h, dk)
            ada(p + "self_attn", False)
        for ffn in ("ffn1", "ffn2"):
            _linear(sd, p + ffn + ".w_1", d, ff, rng)
            _linear(sd, p + ffn + ".w_2", ff, d, rng)
            ada(p + ffn, True)
        ada(p + "conv_module", False)
        sd[p + "conv_module.pointwise_conv1.weight"] = _kaiming(rng, (2 * d, d, 1), d)
        sd[p + "conv_module.pointwise_conv1.bias"] = _kaiming(rng, (2 * d,), 2 * d)
        sd[p + "conv_module.depthwise_conv.weight"] = _kaiming(rng, (d, 1, cnn_module_kernel), cnn_module_kernel)
        sd[p + "conv_module.depthwise_conv.bias"] = _kaiming(rng, (d,), d)
        if cnn_norm_type == "batch_norm":
            _batchnorm(sd, p + "conv_module.norm", d, rng)
        else:
            _layernorm(sd, p + "conv_module.norm", d, rng, perturb_norm)
        sd[p + "conv_module.pointwise_conv2.weight"] = _kaiming(rng, (d, d, 1), d)
        sd[p + "conv_module.pointwise_conv2.bias"] = _kaiming(rng, (d,), d)
        for n in ("layer_norm1", "layer_norm2", "layer_norm3", "layer_norm4"):
            _layernorm(sd, p + n, d, rng, perturb_norm)
    tr_k = 1 if streaming else 5
    sd["encoder.time_reduction_layer.dw_conv.weight"] = _kaiming(rng, (d, 1, tr_k), tr_k)
    sd["encoder.time_reduction_layer.dw_conv.bias"] = _kaiming(rng, (d,), d)
    sd["encoder.time_reduction_layer.pw_conv.weight"] = _kaiming(rng, (d, d, 1), d)
    sd["encoder.time_reduction_layer.pw_conv.bias"] = _kaiming(rng, (d,), d)
    _linear(sd, "encoder.time_recover_layer", d, d, rng)
    if output_size is not None and output_size != d:
        _linear(sd, "encoder.final_proj", d, output_size, rng)
        d = output_size
    sd["ctc.ctc_lo.weight"] = _xavier(rng, (d, vocab_size), d, vocab_size) * np.float32(ctc_sharpen)
    sd["ctc.ctc_lo.bias"] = np.zeros(vocab_size, np.float32)
    return sd


def efficient_conformer_state_dict(stride_layer_idx=3, group_layer_idx=(0, 1, 2, 3), group_size=3, cnn_module_kernel=15,
                                   attention_heads=4, output_size=256, num_blocks=12, seed=1234, **kw):
    """Random-init ``EfficientConformerModel`` inference parameters: the Conformer dict with
    (a) ``pos_bias_u/v`` of shape [h, d_k*group_size] and a ``linear_pos.bias`` on grouped-attention layers
    (efficient_conformer/attention.py:31-37), (b) depthwise kernels of size k//2 after the stride layer
    (efficient_conformer/encoder.py:123-128)."""
    sd = conformer_state_dict(cnn_module_kernel=cnn_module_kernel, attention_heads=attention_heads,
                              output_size=output_size, num_blocks=num_blocks, seed=seed, **kw)
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    d, h = output_size, attention_heads
    dk = d // h
    for i in range(num_blocks):
        p = f"encoder.encoders.{i}."
        if i in tuple(group_layer_idx or ()):
            sd[p + "self_attn.pos_bias_u"] = _xavier(rng, (h, dk * group_size), h, dk * group_size)
            sd[p + "self_attn.pos_bias_v"] = _xavier(rng, (h, dk * group_size), h, dk * group_size)
            sd[p + "self_attn.linear_pos.bias"] = _kaiming(rng, (d,), d)
        strides = [] if stride_layer_idx is None else ([stride_layer_idx] if isinstance(stride_layer_idx, int) else list(stride_layer_idx))
        n_before = sum(1 for v in strides if v < i)
        if n_before:  # cnn_module_kernels: // 2 per stride layer passed (encoder.py:123-128)
            k2 = cnn_module_kernel >> n_before
            sd[p + "conv_module.depthwise_conv.weight"] = _kaiming(rng, (d, 1, k2), k2)
    return sd


def deepspeech2_state_dict(input_dim=80, vocab_size=DEFAULT_VOCAB_SIZE, num_rnn_layers=5, rnn_size=1024, streaming=True,
                           seed=1234, ctc_sharpen=4.0, perturb_norm=False, cmvn_mean=10.0, cmvn_istd=1.0 / 3.3,
                           use_gru=False):
    """Random-init ``DeepSpeech2Model`` inference parameters (deepspeech2/encoder.py:8-55): plain ``nn.Conv2D``
    (fan-in uniform), ``nn.LSTM`` / ``nn.GRU`` when ``use_gru`` (U(+-1/sqrt(H)), Paddle default; 4H / 3H gate rows),
    ``nn.LayerNorm``, CTC ``nn.Linear``."""
    rng = np.random.Generator(np.random.PCG64(seed))
    H, D = rnn_size, (1 if streaming else 2)
    G = 3 if use_gru else 4
    f2 = ((input_dim - 1) // 2 - 1) // 2
    sd = {}
    sd["encoder.global_cmvn.mean"] = np.full(input_dim, cmvn_mean, np.float32)
    sd["encoder.global_cmvn.istd"] = np.full(input_dim, cmvn_istd, np.float32)
    sd["encoder.conv.conv.0.weight"] = _kaiming(rng, (32, 1, 3, 3), 9)
    sd["encoder.conv.conv.0.bias"] = _kaiming(rng, (32,), 9)
    sd["encoder.conv.conv.2.weight"] = _kaiming(rng, (32, 32, 3, 3), 32 * 9)
    sd["encoder.conv.conv.2.bias"] = _kaiming(rng, (32,), 32 * 9)
    for l in range(num_rnn_layers):
        in_dim = 32 * f2 if l == 0 else D * H
        for d in range(D):
            sfx = "_l0" if d == 0 else "_l0_reverse"
            p = f"encoder.rnn.{l}."
            sd[p + "weight_ih" + sfx] = _kaiming(rng, (G * H, in_dim), H)
            sd[p + "weight_hh" + sfx] = _kaiming(rng, (G * H, H), H)
            sd[p + "bias_ih" + sfx] = _kaiming(rng, (G * H,), H)
            sd[p + "bias_hh" + sfx] = _kaiming(rng, (G * H,), H)
        _layernorm(sd, f"encoder.layernorm_list.{l}", D * H, rng, perturb_norm)
    sd["decoder.ctc_lo.weight"] = _xavier(rng, (D * H, vocab_size), D * H, vocab_size) * np.float32(ctc_sharpen)
    sd["decoder.ctc_lo.bias"] = np.zeros(vocab_size, np.float32)
    return sd
