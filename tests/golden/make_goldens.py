"""Regenerate the committed golden vectors.  Run in the BUILD container (needs /root/reference):

    python tests/golden/make_goldens.py

1. ctc_greedy_golden.npz  -- inputs + outputs of the REFERENCE's own
   ppasr/decoders/ctc_greedy_decoder.py (imported unmodified from /root/reference; numpy-only).
   This pins oracle/ctc_decoders_oracle.py and the HIP greedy kernels to the real reference.
2. conformer_oracle_golden.npz -- outputs of oracle/conformer_oracle.py on seeded weights/inputs.
   PaddlePaddle cannot be imported here, so this only pins the oracle against drift
   ("parity unpinned" for the encoder, see oracle/conformer_oracle.py header).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def greedy_cases():
    rng = np.random.Generator(np.random.PCG64(42))
    cases = []
    V = 37
    for T in (1, 2, 5, 16, 64, 249):
        for kind in ("random", "peaky", "blanky", "ties"):
            p = rng.random((T, V)).astype(np.float32)
            if kind == "peaky":  # long runs of the same symbol, some blanks
                idx = np.repeat(rng.integers(0, V, size=(T + 3) // 4), 4)[:T]
                p[np.arange(T), idx] += 2.0
            elif kind == "blanky":  # mostly blank
                p[:, 0] += np.where(rng.random(T) < 0.8, 3.0, 0.0).astype(np.float32)
            elif kind == "ties":  # exact ties -> first index must win
                p = np.round(p * 4).astype(np.float32) / 4
            p = p / p.sum(axis=1, keepdims=True)
            cases.append(p.astype(np.float32))
    cases.append(np.tile(np.eye(V, dtype=np.float32)[0], (9, 1)))  # all blank -> empty text, score 0
    return cases, V


def main():
    sys.path.insert(0, "/root/reference")
    from ppasr.decoders.ctc_greedy_decoder import greedy_decoder, greedy_decoder_batch, greedy_decoder_chunk

    cases, V = greedy_cases()
    vocab = ["<blank>"] + [chr(0x4E00 + i) for i in range(V - 2)] + ["<space>"]
    out = {"vocab_size": np.int64(V), "n_cases": np.int64(len(cases))}
    for i, p in enumerate(cases):
        score, text = greedy_decoder(p, vocab)
        out[f"probs_{i}"] = p
        out[f"score_{i}"] = np.float64(score)
        out[f"text_{i}"] = np.array(text)
    same_len = [p for p in cases if p.shape[0] == 64]
    out["batch_texts"] = np.array(greedy_decoder_batch(same_len, vocab))
    out["batch_ids"] = np.array([i for i, p in enumerate(cases) if p.shape[0] == 64], np.int64)
    # streaming decoder: feed case (T=249, peaky) in chunks of 16
    big = [i for i, p in enumerate(cases) if p.shape[0] == 249][1]
    l1, l2, texts, scores = None, None, [], []
    for s in range(0, 249, 16):
        score, text, l1, l2 = greedy_decoder_chunk(cases[big][s:s + 16], vocab, l1, l2)
        texts.append(text)
        scores.append(score)
    out["chunk_case"] = np.int64(big)
    out["chunk_texts"] = np.array(texts)
    out["chunk_scores"] = np.array(scores, np.float64)
    np.savez_compressed(os.path.join(HERE, "ctc_greedy_golden.npz"), **out)
    print("wrote ctc_greedy_golden.npz:", len(cases), "cases")

    import torch
    from oracle.conformer_oracle import ConformerOracle
    from ppasr_amd.utils.synth import conformer_state_dict, synth_features
    torch.set_num_threads(1)  # deterministic reduction order
    V2, L = 64, 2
    sd = conformer_state_dict(vocab_size=V2, num_blocks=L, seed=77, perturb_norm=True)
    x, lens = synth_features(2, 99, lens=[99, 61], seed=78)
    o = ConformerOracle(sd, num_blocks=L)
    probs, logits = o.get_encoder_out(x, lens, return_logits=True)
    # streaming: the first utterance in two chunks (67-frame window, stride 64 -> 16 frames + rest)
    p1, att, cnn = o.get_encoder_out_chunk(x[:1, :67], 0, -1)
    p2, att2, cnn2 = o.get_encoder_out_chunk(x[:1, 64:99], 16, -1, att, cnn)
    np.savez_compressed(os.path.join(HERE, "conformer_oracle_golden.npz"),
                        logits=logits.numpy(), probs_argmax=probs.argmax(-1).numpy(),
                        chunk1_probs=p1.numpy(), chunk2_probs=p2.numpy(), att_cache_shape=np.array(att2.shape),
                        cnn_cache=cnn2.numpy())
    print("wrote conformer_oracle_golden.npz", tuple(logits.shape), tuple(p1.shape), tuple(p2.shape))


if __name__ == "__main__":
    main()
