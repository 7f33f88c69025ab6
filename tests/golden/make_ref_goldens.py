"""Regenerate tests/golden/ref_small.npz (and, with --full, ref_full.npz) by running the REFERENCE's OWN model source.

    python tests/golden/make_ref_goldens.py [--full] [--only NAME ...]

How: `sys.path[:0] = [oracle/paddle_shim, /root/reference]` -- the torch-CPU-backed `paddle` of oracle/paddle_shim lets
`/root/reference/ppasr/model_utils/{conformer,efficient_conformer,squeezeformer,deepspeech2}/model.py` import and run
UNMODIFIED (oracle/paddle_shim/README.md lists the Paddle operator semantics the shim itself supplies).  For every case
of tests/ref_cases.py the reference model is built the way `PPASRTrainer.__setup_model` does (trainer.py:172-210), a
seeded state dict is loaded BY NAME (every synthetic name must exist in the model's own `state_dict()` with the same
shape -- that is the check of the Paddle parameter names / layouts the importer relies on), and
`get_encoder_out` / `get_encoder_out_chunk` are called the way `trainer.py:626` / `inference_predictor.py:147-212` call
them.  Needs /root/reference, so it runs in the build container only; the `.npz` files it writes travel.

Stored per batched case:  `<name>/probs` (= get_encoder_out), `<name>/logits` (= ctc_lo(encoder(...)), the pre-softmax
activations of the same reference modules).  Per streaming case and `required_cache_size`: the concatenated chunk
outputs, the chunk lengths and the final caches / states.  Full-size configs: greedy ids and top-2 margins of every
frame, logits on a fixed subset of vocabulary columns, log-sum-exp per frame (ref_cases.sampled_columns), and for the
beam-search configs the token sequences of oracle/ctc_beam_search_oracle.c run on the REFERENCE's probabilities.
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_cases as rc  # noqa: E402


def enter_reference():
    sys.path[:0] = [os.path.join(ROOT, "oracle", "paddle_shim"), REFERENCE]
    import paddle  # noqa: F401  (the shim)
    import torch
    torch.set_grad_enabled(False)
    return paddle


def build_reference_model(paddle, case, sd):
    """Construct the reference model like trainer.py:172-210 and load `sd` by name."""
    from ppasr.model_utils.conformer.model import ConformerModel
    from ppasr.model_utils.deepspeech2.model import DeepSpeech2Model
    from ppasr.model_utils.efficient_conformer.model import EfficientConformerModel
    from ppasr.model_utils.squeezeformer.model import SqueezeformerModel
    fam = case["family"]
    cls = {"conformer": ConformerModel, "efficient_conformer": EfficientConformerModel,
           "squeezeformer": SqueezeformerModel, "deepspeech2": DeepSpeech2Model}[fam]
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump({"mean": sd["encoder.global_cmvn.mean"].tolist(), "istd": sd["encoder.global_cmvn.istd"].tolist()}, f)
        mean_istd = f.name
    enc = rc.reference_encoder_conf(case)
    if fam == "deepspeech2":
        model = cls(input_dim=80, vocab_size=case["V"], mean_istd_path=mean_istd, streaming=case["streaming"],
                    encoder_conf=enc, decoder_conf=dict(dropout_rate=0.1))
    else:
        dec = dict(attention_heads=4, linear_units=1024, num_blocks=3, r_num_blocks=3, dropout_rate=0.1,
                   positional_dropout_rate=0.1, self_attention_dropout_rate=0.1, src_attention_dropout_rate=0.1)
        model = cls(input_dim=80, vocab_size=case["V"], mean_istd_path=mean_istd, streaming=case["streaming"],
                    encoder_conf=enc, decoder_conf=dec, ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3,
                    length_normalized_loss=False)
    os.unlink(mean_istd)
    own = model.state_dict()
    for k, v in sd.items():
        assert k in own, f"{fam}: synthetic parameter {k} is not a name of the reference model"
        assert list(own[k].shape) == list(v.shape), f"{fam}: {k} shape {v.shape} vs reference {own[k].shape}"
    missing, unexpected = model.set_state_dict(sd)
    assert not unexpected, unexpected
    # what the synthetic dict does not carry: the attention decoder (training only), the RNN cell aliases, and the
    # stride layer's `concat_linear`, which StrideConformerEncoderLayer creates unconditionally and never uses with
    # concat_after=False (efficient_conformer/encoder.py:447,497-501)
    stray = [m for m in missing if not (m.startswith("decoder.") and fam != "deepspeech2") and ".cell" not in m
             and ".concat_linear." not in m]
    assert not stray, f"{fam}: reference parameters without a synthetic value: {stray[:8]}"
    model.eval()
    return model


def run_batched(paddle, model, case, x, lens):
    xs = paddle.to_tensor(x)
    ls = paddle.to_tensor(lens, dtype=paddle.int64)
    probs = model.get_encoder_out(xs, ls)
    if case["family"] == "deepspeech2":
        eouts, _, _, _ = model.encoder(xs, ls, None, None)
        logits = model.decoder.ctc_lo(eouts)
    else:
        enc, _ = model.encoder(xs, ls, decoding_chunk_size=-1, num_decoding_left_chunks=-1)
        logits = model.ctc.ctc_lo(enc)
    return probs.numpy(), logits.numpy()


def run_chunks_former(paddle, model, x, required):
    """InferencePredictor.predict_chunk_conformer state machine (inference_predictor.py:184-220)."""
    att = paddle.zeros([0, 0, 0, 0])
    cnn = paddle.zeros([0, 0, 0, 0])
    offset = 0
    outs = []
    for (a, b) in rc.windows(x.shape[1]):
        p, att, cnn = model.get_encoder_out_chunk(paddle.to_tensor(x[:, a:b]), offset, required, att, cnn)
        outs.append(p.numpy())
        offset += p.shape[1]
    return (np.concatenate(outs, 1), np.array([o.shape[1] for o in outs], np.int32), att.numpy(), cnn.numpy())


def run_chunks_ds2(paddle, model, case, x):
    """InferencePredictor.predict_chunk_deepspeech (inference_predictor.py:147-182): zero initial boxes, states carried."""
    B = x.shape[0]
    h = paddle.zeros([case["L"], B, 1024])
    c = paddle.zeros([case["L"], B, 1024])
    outs, lens_out = [], []
    for (a, b) in rc.windows(x.shape[1]):
        xl = paddle.to_tensor(np.full(B, b - a, np.int64))
        p, ln, h, c = model.get_encoder_out_chunk(paddle.to_tensor(x[:, a:b]), xl, h, c)
        outs.append(p.numpy())
        lens_out.append(ln.numpy())
    c_np = c.numpy() if c is not None else np.zeros((0,), np.float32)
    return np.concatenate(outs, 1), np.array([o.shape[1] for o in outs], np.int32), h.numpy(), c_np


def small(paddle, only):
    out = {}
    for name, case in rc.SMALL.items():
        if only and name not in only:
            continue
        t0 = time.time()
        sd = rc.state_dict(case)
        model = build_reference_model(paddle, case, sd)
        x, lens = rc.features(case)
        probs, logits = run_batched(paddle, model, case, x, lens)
        out[f"{name}/probs"], out[f"{name}/logits"] = probs, logits
        if case["chunk_frames"]:
            xc = rc.chunk_features(case)
            if case["family"] == "deepspeech2":
                p, n, h, c = run_chunks_ds2(paddle, model, case, xc)
                out[f"{name}/chunk/probs"], out[f"{name}/chunk/n"] = p, n
                out[f"{name}/chunk/h"], out[f"{name}/chunk/c"] = h, c
            else:
                for req in case["required"]:
                    p, n, att, cnn = run_chunks_former(paddle, model, xc, req)
                    k = f"{name}/chunk{req}"
                    out[k + "/probs"], out[k + "/n"], out[k + "/att"], out[k + "/cnn"] = p, n, att, cnn
        print(f"[ref] {name}: probs {probs.shape}  ({time.time() - t0:.1f} s)", flush=True)
    return out


def _beam_oracle():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libctc_beam_oracle.so"))
    lib.ctc_beam_oracle_decode.restype = ctypes.c_int
    return lib


def beam_tokens(lib, p, n_frames):
    """oracle/ctc_beam_search_oracle.c on one utterance's probabilities [T', V] -> best token sequence."""
    T, V = p.shape
    tokens = np.full((1, T), -1, np.int32)
    lens = np.empty(1, np.int32)
    scores = np.empty(1, np.float64)
    p = np.ascontiguousarray(p[:n_frames], np.float32)
    rc_ = lib.ctc_beam_oracle_decode(p.ctypes.data_as(ctypes.c_void_p), n_frames, V, rc.BEAM["beam_size"],
                                     ctypes.c_double(rc.BEAM["cutoff_prob"]), rc.BEAM["cutoff_top_n"], 0, 1, T,
                                     tokens.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p),
                                     scores.ctypes.data_as(ctypes.c_void_p))
    assert rc_ >= 1  # number of hypotheses returned
    return tokens[0, :int(lens[0])].copy(), float(scores[0])


def summarise(logits, cols):
    """[B, T', V] logits -> per-frame greedy id, top-2 margin, log-sum-exp, and the logits of `cols`."""
    l64 = logits.astype(np.float64)
    ids = l64.argmax(-1).astype(np.int32)
    part = np.partition(l64, -2, axis=-1)
    margin = (part[..., -1] - part[..., -2]).astype(np.float32)
    m = l64.max(-1, keepdims=True)
    lse = (m[..., 0] + np.log(np.exp(l64 - m).sum(-1))).astype(np.float32)
    return ids, margin, lse, np.ascontiguousarray(logits[..., cols])


def full(paddle, only):
    out = {}
    lib = _beam_oracle()
    for name, case in rc.FULL.items():
        if only and name not in only:
            continue
        t0 = time.time()
        sd = rc.state_dict(case)
        model = build_reference_model(paddle, case, sd)
        x, lens = rc.features(case)
        cols = rc.sampled_columns(case["V"])
        if name == "cfg5":
            # length buckets (200-frame width), each padded to its longest member: what the bucketed route computes
            buckets = {}
            for i, ln in enumerate(lens):
                buckets.setdefault(rc.bucket_of(ln), []).append(i)
            toks, nt = np.full((len(lens), 800), -1, np.int32), np.zeros(len(lens), np.int32)
            for bk, idx in sorted(buckets.items()):
                Tb = int(lens[idx].max())
                probs, logits = run_batched(paddle, model, case, x[idx, :Tb], lens[idx])
                ids, margin, lse, samp = summarise(logits, cols)
                for j, i in enumerate(idx):
                    nf = (int(lens[i]) + 3) // 4  # valid output frames (mask slicing of subsampling.py:115 = ceil(len/4))
                    nf = min(nf, probs.shape[1])
                    t, _ = beam_tokens(lib, probs[j], nf)
                    toks[i, :len(t)], nt[i] = t, len(t)
                    out[f"{name}/ids/{i}"], out[f"{name}/margin/{i}"] = ids[j, :nf], margin[j, :nf]
                    out[f"{name}/lse/{i}"], out[f"{name}/sampled/{i}"] = lse[j, :nf], samp[j, :nf]
            out[f"{name}/beam_tokens"], out[f"{name}/beam_n"] = toks, nt
            # the same 16 utterances as ONE batch padded to the longest: the reference's values depend on the batch an
            # utterance is padded into (full-context attention sees the partially padded last frame, every PAD row is
            # computed), so the one-batch route has its own reference
            probs, logits = run_batched(paddle, model, case, x, lens)
            ids, margin, lse, samp = summarise(logits, cols)
            for i in range(len(lens)):
                nf = min((int(lens[i]) + 3) // 4, probs.shape[1])
                out[f"{name}pad/ids/{i}"], out[f"{name}pad/margin/{i}"] = ids[i, :nf], margin[i, :nf]
                out[f"{name}pad/lse/{i}"], out[f"{name}pad/sampled/{i}"] = lse[i, :nf], samp[i, :nf]
        else:
            probs, logits = run_batched(paddle, model, case, x, lens)
            ids, margin, lse, samp = summarise(logits, cols)
            out[f"{name}/ids"], out[f"{name}/margin"], out[f"{name}/lse"], out[f"{name}/sampled"] = ids, margin, lse, samp
            out[f"{name}/maxprob"] = probs.max(-1).astype(np.float32)
            if name == "cfg4":
                B, Tp, _ = probs.shape
                toks, nt = np.full((B, Tp), -1, np.int32), np.zeros(B, np.int32)
                for b in range(B):
                    t, _ = beam_tokens(lib, probs[b], Tp)
                    toks[b, :len(t)], nt[b] = t, len(t)
                out[f"{name}/beam_tokens"], out[f"{name}/beam_n"] = toks, nt
        out[f"{name}/cols"] = cols
        print(f"[ref] {name}: done ({time.time() - t0:.1f} s)", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also regenerate ref_full.npz (BASELINE configs, minutes of CPU)")
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--add", nargs="*", default=None,
                    help="run only these SMALL cases and MERGE their outputs into the existing ref_small.npz")
    ap.add_argument("--add-full", nargs="*", default=None,
                    help="run only these FULL cases and MERGE their outputs into the existing ref_full.npz")
    args = ap.parse_args()
    if not os.path.isdir(REFERENCE):
        raise SystemExit("needs /root/reference (build container only)")
    paddle = enter_reference()
    if args.add:
        path = os.path.join(HERE, "ref_small.npz")
        old = dict(np.load(path))
        new = small(paddle, args.add)
        old = {k: v for k, v in old.items() if k.split("/")[0] not in args.add}
        old.update(new)
        np.savez_compressed(path, **old)
        print("merged", sorted({k.split("/")[0] for k in new}), "into ref_small.npz:", len(old), "arrays")
        return
    if args.add_full:
        path = os.path.join(HERE, "ref_full.npz")
        old = dict(np.load(path))
        new = full(paddle, args.add_full)
        old = {k: v for k, v in old.items() if k.split("/")[0] not in args.add_full}
        old.update(new)
        np.savez_compressed(path, **old)
        print("merged", sorted({k.split("/")[0] for k in new}), "into ref_full.npz:", len(old), "arrays")
        return
    if not args.full or args.only:
        out = small(paddle, args.only)
        if out and not args.only:
            np.savez_compressed(os.path.join(HERE, "ref_small.npz"), **out)
            print("wrote ref_small.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")
    if args.full:
        out = full(paddle, args.only)
        if out and not args.only:
            np.savez_compressed(os.path.join(HERE, "ref_full.npz"), **out)
            print("wrote ref_full.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
