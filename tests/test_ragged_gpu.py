"""Ragged batches with ``set_skip_padding`` (ppasr_set_skip_padding): per utterance only the rows its valid output
frames depend on are computed.  The valid rows must be BIT-identical to the default mode (which the other test files
hold against the oracles), everything behind them zero / blank, and the greedy tokens with length trimming equal."""
import numpy as np
import pytest
import torch

from ppasr_amd.utils.synth import (conformer_state_dict, efficient_conformer_state_dict, squeezeformer_state_dict,
                                   synth_features)

pytestmark = pytest.mark.gpu


def _conformer(V, streaming):
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    L = 3
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=5, perturb_norm=True)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
    return ConformerModel(80, V, streaming=streaming, encoder_conf=conf, state_dict=sd, device="cuda:0"), 4


def _squeezeformer(V):
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    L = 4
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=6, perturb_norm=True)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=1, recover_idx=3,
                feed_forward_expansion_factor=8, cnn_module_kernel=31)
    return SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0"), 4


def _efficient(V):
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    L = 4
    sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=7, perturb_norm=True, stride_layer_idx=1,
                                        group_layer_idx=(0, 1))
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                cnn_module_norm="layer_norm",
                efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3,
                                    stride_kernel=True))
    return EfficientConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0"), 8


def _efficient_noncausal(V):
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    L = 4
    sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=8, perturb_norm=True, stride_layer_idx=1,
                                        group_layer_idx=(0, 1))
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                cnn_module_norm="layer_norm",
                efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3,
                                    stride_kernel=True))
    return EfficientConformerModel(80, V, streaming=False, encoder_conf=conf, state_dict=sd, device="cuda:0"), 8


def _squeezeformer_noncausal(V):
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    L = 4
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=9, perturb_norm=True, streaming=False)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=L, reduce_idx=1, recover_idx=3,
                feed_forward_expansion_factor=8, cnn_module_kernel=31)
    return SqueezeformerModel(80, V, streaming=False, encoder_conf=conf, state_dict=sd, device="cuda:0"), 4


def _conformer_front(V, input_layer, mul):
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    L = 2
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=10 + mul, perturb_norm=True, input_layer=input_layer)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15, input_layer=input_layer)
    return ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0"), mul


def _conformer512(V, streaming, **opts):
    """output_size 512 / 8 heads (+ constructor options): the general layer route (csrc/capi_generic.hip)"""
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    L = 2
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=21, perturb_norm=True, output_size=512, attention_heads=8,
                              **{k: v for k, v in opts.items() if k not in ("activation_type", "normalize_before")})
    conf = dict(output_size=512, attention_heads=8, linear_units=2048, num_blocks=L, cnn_module_kernel=15, **opts)
    return ConformerModel(80, V, streaming=streaming, encoder_conf=conf, state_dict=sd, device="cuda:0"), 4


def _squeezeformer512(V):
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    L = 3
    sd = squeezeformer_state_dict(vocab_size=V, num_blocks=L, seed=22, perturb_norm=True, encoder_dim=512, attention_heads=8)
    conf = dict(encoder_dim=512, output_size=512, attention_heads=8, num_blocks=L, reduce_idx=1, recover_idx=2,
                feed_forward_expansion_factor=8, cnn_module_kernel=31)
    return SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0"), 4


def _efficient512(V, streaming):
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    L = 3
    sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=L, seed=23, perturb_norm=True, stride_layer_idx=1,
                                        group_layer_idx=(0, 1), output_size=512, attention_heads=8)
    conf = dict(output_size=512, attention_heads=8, linear_units=2048, num_blocks=L, cnn_module_kernel=15,
                cnn_module_norm="layer_norm",
                efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3,
                                    stride_kernel=True))
    return EfficientConformerModel(80, V, streaming=streaming, encoder_conf=conf, state_dict=sd, device="cuda:0"), 8


FAMILIES = {
    "conformer": lambda V: _conformer(V, True),
    # the general layer route (width 512; a post-norm / no-macaron / abs_pos variant; the other two families)
    "conformer512": lambda V: _conformer512(V, True),
    "conformer512-noncausal": lambda V: _conformer512(V, False),
    "conformer512-postnorm-abs": lambda V: _conformer512(V, True, normalize_before=False, macaron_style=False,
                                                         pos_enc_layer_type="abs_pos", activation_type="relu"),
    "squeezeformer512": _squeezeformer512,
    "efficient512": lambda V: _efficient512(V, True),
    "efficient512-noncausal": lambda V: _efficient512(V, False),
    # the 6x / 8x front ends (input_layer: conv2d6 / conv2d8): the layers skip, the front end computes every row
    "conformer-conv2d6": lambda V: _conformer_front(V, "conv2d6", 6),
    "conformer-conv2d8": lambda V: _conformer_front(V, "conv2d8", 8),
    "conformer-noncausal": lambda V: _conformer(V, False),
    "squeezeformer": _squeezeformer,
    "efficient": _efficient,
    "efficient-noncausal": _efficient_noncausal,
    "squeezeformer-noncausal": _squeezeformer_noncausal,
}


@pytest.mark.parametrize("family", list(FAMILIES))
@pytest.mark.parametrize("lens", [[900, 611, 420, 133, 36, 7], [1203, 1203, 300], [260, 258, 257, 131, 129, 128, 127, 1]])
@pytest.mark.parametrize("route", [-1, 0])  # default route selection / always the fused kernels (ppasr_set_ffn_split)
def test_skip_padding_equals_default_on_valid_rows(family, lens, route):
    V = 211
    model, mul = FAMILIES[family](V)
    model.set_ffn_split(route)
    B, T = len(lens), max(lens)
    x, lens_a = synth_features(B, T, lens=lens, seed=T + B)
    p0, l0 = model.get_encoder_out(x, lens_a, return_logits=True)
    t0, n0, s0 = model.encode_greedy(x, lens_a, trim_to_length=True)
    model.set_skip_padding(True)
    try:
        p1, l1 = model.get_encoder_out(x, lens_a, return_logits=True)
        t1, n1, s1 = model.encode_greedy(x, lens_a, trim_to_length=True)
    finally:
        model.set_skip_padding(False)
    torch.cuda.synchronize()
    Tp = p0.shape[1]
    for b, ln in enumerate(lens):
        nv = min(Tp, (ln + mul - 1) // mul)
        assert torch.equal(p0[b, :nv], p1[b, :nv]), (family, b, ln)
        assert torch.equal(l0[b, :nv], l1[b, :nv]), (family, b, ln)
        assert not bool(p1[b, nv:].any()) and not bool(l1[b, nv:].any()), (family, b, ln)
    assert bool(torch.isfinite(p1).all())
    assert torch.equal(n0, n1) and torch.equal(t0, t1)
    assert np.allclose(s0.cpu().numpy(), s1.cpu().numpy(), rtol=0, atol=0)
    # trim_to_length decodes exactly the valid frames (mul * t < len; mul = 8 behind the Efficient-Conformer's stride
    # layer): the tokens are the collapse of the first nv frames' argmax, no PAD frame contributes
    ids = p0.argmax(-1).cpu().numpy()
    for b, ln in enumerate(lens):
        nv = min(Tp, (ln + mul - 1) // mul)
        fr = ids[b, :nv]
        keep = (np.concatenate([[True], fr[1:] != fr[:-1]]) if nv else np.zeros(0, bool)) & (fr != 0)
        assert np.array_equal(fr[keep], t0[b, :int(n0[b])].cpu().numpy()), (family, b, ln)


def test_skip_padding_without_lengths_is_the_default_mode():
    model, _ = _conformer(97, True)
    x, lens = synth_features(2, 200, seed=3)
    p0 = model.get_encoder_out(x, lens)
    model.set_skip_padding(True)
    p1 = model.get_encoder_out(x, lens)  # full-length utterances: nothing to skip
    model.set_skip_padding(False)
    assert torch.equal(p0, p1)


@pytest.mark.parametrize("decoder", ["ctc_greedy", "ctc_beam_search"])
def test_evaluate_trim_padding(decoder):
    """evaluate(trim_padding=True) on a ragged batch = decoding, per utterance, only the rows the reference's own
    length mask calls valid (4t < len) of the default-mode probabilities; the reference's default additionally decodes
    the padded rows of the shorter utterances."""
    from ppasr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decoder_batch
    from ppasr_amd.evaluate import evaluate
    from ppasr_amd.utils.metrics import cer, labels_to_string
    from ppasr_amd.utils.synth import synth_vocabulary
    V = 150
    vocab = synth_vocabulary(V)
    model, _ = _conformer(V, True)
    bsd = BeamSearchDecoder(0.0, 0.0, 8, 0.99, 40, vocab) if decoder == "ctc_beam_search" else None
    lens = [403, 251, 120, 64]
    x, la = synth_features(len(lens), max(lens), lens=lens, seed=11)
    rng = np.random.Generator(np.random.PCG64(5))
    labels = rng.integers(2, V - 1, size=(len(lens), 30)).astype(np.int64)
    probs = model.get_encoder_out(x, la)
    nv = model.valid_out_frames(la, x.shape[1]).cpu().tolist()
    assert nv == [min(probs.shape[1], (ln + 3) // 4) for ln in lens]
    want = 0.0
    for b in range(len(lens)):
        one = probs[b:b + 1, :nv[b]].contiguous()
        text = bsd.decode_batch_beam_search_offline(one)[0] if bsd else greedy_decoder_batch(one, vocab)[0]
        want += cer(text, labels_to_string(labels[b:b + 1], vocab, eos=V - 1)[0])
    got = evaluate(model, [(x, labels, la, None)], vocab, decoder=decoder, beam_search_decoder=bsd, trim_padding=True)
    assert got == pytest.approx(want / len(lens))
    assert not model.__dict__.get("skip_padding", False)  # evaluate restores the default mode
