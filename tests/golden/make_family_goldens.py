"""Regenerate tests/golden/family_oracle_golden.npz: outputs of the CPU restatements of the OTHER paths (Squeezeformer,
Efficient-Conformer, DeepSpeech2, the C prefix beam search, the fbank front-end) on seeded inputs.

    python tests/golden/make_family_goldens.py

None of these can be produced by the reference itself here (PaddlePaddle, paddleaudio and paddlespeech_ctcdecoders are
not installable offline), so -- like conformer_oracle_golden.npz -- they pin the ORACLES against drift and give the GPU
tests fixtures that do not need an oracle at run time; the paths stay "parity unpinned" against the real reference.
The inputs are regenerated from seeds by `cases()` below (shared with the tests), only outputs are stored."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "family_oracle_golden.npz")


def beam_probs(seed, T, V):
    rng = np.random.Generator(np.random.PCG64(seed))
    logits = rng.standard_normal((T, V)).astype(np.float32)
    idx = np.repeat(rng.integers(0, V, size=(T + 2) // 3), 3)[:T]
    logits[np.arange(T), idx] += 5.0
    logits[:, 0] += np.where(rng.random(T) < 0.4, 6.0, 0.0).astype(np.float32)
    e = np.exp(logits - logits.max(1, keepdims=True))
    return (e / e.sum(1, keepdims=True)).astype(np.float32)


def audio(seed, seconds, sr=16000):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(int(sr * seconds)) / sr
    x = 0.2 * np.sin(2 * np.pi * 180 * t) * (1 + 0.5 * np.sin(2 * np.pi * 2.5 * t)) + 0.05 * rng.standard_normal(t.shape)
    return x.astype(np.float32)


def cases():
    """-> dict of seeded models / inputs used by the generator and by the tests."""
    from ppasr_amd.utils.synth import (deepspeech2_state_dict, efficient_conformer_state_dict, squeezeformer_state_dict,
                                       synth_features)
    c = {}
    c["sq_sd"] = squeezeformer_state_dict(vocab_size=61, num_blocks=4, seed=301, perturb_norm=True)
    c["sq_conf"] = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=4, reduce_idx=1, recover_idx=3,
                        feed_forward_expansion_factor=8, cnn_module_kernel=31)
    c["sq_x"], c["sq_lens"] = synth_features(2, 131, lens=[131, 77], seed=302)
    c["eff_sd"] = efficient_conformer_state_dict(vocab_size=53, num_blocks=4, seed=303, perturb_norm=True, stride_layer_idx=1,
                                                 group_layer_idx=(0, 1))
    c["eff_conf"] = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=4, cnn_module_kernel=15,
                         cnn_module_norm="layer_norm",
                         efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3,
                                             stride_kernel=True))
    c["eff_x"], c["eff_lens"] = synth_features(2, 147, lens=[147, 90], seed=304)
    for streaming in (True, False):
        k = "ds2s" if streaming else "ds2b"
        c[k + "_sd"] = deepspeech2_state_dict(vocab_size=47, num_rnn_layers=2, streaming=streaming, seed=305 + streaming,
                                              perturb_norm=True)
        c[k + "_x"], c[k + "_lens"] = synth_features(3, 99, lens=[99, 64, 31], seed=307)
    c["beam"] = [(beam_probs(311, 60, 90), 10, 0.99, 40), (beam_probs(312, 35, 40), 25, 1.0, 40),
                 (beam_probs(313, 80, 300), 50, 0.99, 20)]
    c["wav"] = audio(321, 0.63)
    return c


def beam_oracle():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libctc_beam_oracle.so"))
    lib.ctc_beam_oracle_decode.restype = ctypes.c_int
    return lib


def beam_decode(lib, p, beam, cutoff_prob, top_n):
    T, V = p.shape
    tokens = np.full((1, T), -1, np.int32)
    lens = np.empty(1, np.int32)
    scores = np.empty(1, np.float64)
    p = np.ascontiguousarray(p, np.float32)
    lib.ctc_beam_oracle_decode(p.ctypes.data_as(ctypes.c_void_p), T, V, beam, ctypes.c_double(cutoff_prob), top_n, 0, 1, T,
                               tokens.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p),
                               scores.ctypes.data_as(ctypes.c_void_p))
    return tokens[0, :lens[0]].copy(), float(scores[0])


def compute(c):
    import torch
    from oracle import fbank_oracle
    from oracle.deepspeech2_oracle import DeepSpeech2Oracle
    from oracle.efficient_conformer_oracle import EfficientConformerOracle
    from oracle.squeezeformer_oracle import SqueezeformerOracle
    torch.set_num_threads(1)
    out = {}
    _, lg = SqueezeformerOracle(c["sq_sd"], num_blocks=4, reduce_idx=1, recover_idx=3).get_encoder_out(
        c["sq_x"], c["sq_lens"], return_logits=True)
    out["sq_logits"] = lg.numpy()
    _, lg = EfficientConformerOracle(c["eff_sd"], num_blocks=4, stride_layer_idx=1, group_layer_idx=(0, 1)).get_encoder_out(
        c["eff_x"], c["eff_lens"], return_logits=True)
    out["eff_logits"] = lg.numpy()
    for k, streaming in (("ds2s", True), ("ds2b", False)):
        probs, lens, h, cc = DeepSpeech2Oracle(c[k + "_sd"], 2, 1024, streaming).forward(c[k + "_x"], c[k + "_lens"])
        out[k + "_probs"] = np.asarray(probs, np.float32)
        out[k + "_lens"] = np.asarray(lens, np.int64)
        out[k + "_h"] = np.asarray(h, np.float32)
    lib = beam_oracle()
    for i, (p, beam, cp, tn) in enumerate(c["beam"]):
        tok, sc = beam_decode(lib, p, beam, cp, tn)
        out[f"beam{i}_tokens"] = tok
        out[f"beam{i}_score"] = np.float64(sc)
    out["fbank"] = fbank_oracle.featurize(c["wav"], 16000, 80, True, -20.0).astype(np.float32)
    return out


if __name__ == "__main__":
    out = compute(cases())
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: getattr(v, "shape", None) for k, v in out.items()})
