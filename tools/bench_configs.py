"""Timings of the other BASELINE.json configs on ONE GPU (parity-test cases, not the headline bench line):
  cfg1 DeepSpeech2 non-streaming (bidirectional) B=1, 5 s, greedy
  cfg4 Efficient-Conformer streaming B=64, 10 s, ctc_beam_search beam=10 (cutoff 0.99 / top-40, no LM)
  cfg5 Squeezeformer streaming, 16 utterances/GPU with lengths U{200..3000} frames in length buckets, beam search
Prints one JSON line per config."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decode_ids
from ppasr_amd.utils.synth import (deepspeech2_state_dict, efficient_conformer_state_dict, squeezeformer_state_dict,
                                   synth_features)
V = 4233


def timeit(fn, steps=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps


def cfg1():
    from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
    sd = deepspeech2_state_dict(vocab_size=V, streaming=False, seed=1234)
    m = DeepSpeech2Model(80, V, streaming=False, encoder_conf=dict(num_rnn_layers=5, rnn_size=1024), state_dict=sd)
    x, lens = synth_features(1, 498, seed=20340)
    x, lens = torch.from_numpy(x).cuda(), torch.from_numpy(lens).cuda()
    dt = timeit(lambda: greedy_decode_ids(m.get_encoder_out(x, lens)), 5, 1)
    return {"config": "cfg1 DeepSpeech2 bidirectional B=1 5s greedy", "ms": round(dt * 1e3, 2),
            "audio_s_per_s": round(4.98 / dt, 1)}


def cfg4():
    from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
    sd = efficient_conformer_state_dict(vocab_size=V, seed=1234)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12, cnn_module_kernel=15, cnn_module_norm="layer_norm",
                efficient_conf=dict(stride_layer_idx=[3], stride=[2], group_layer_idx=[0, 1, 2, 3], group_size=3))
    m = EfficientConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd)
    x, lens = synth_features(64, 1000, seed=20640)
    x, lens = torch.from_numpy(x).cuda(), torch.from_numpy(lens).cuda()
    enc = timeit(lambda: m.get_encoder_out(x, lens), 5, 1)
    probs = m.get_encoder_out(x, lens)
    beam = timeit(lambda: beam_search_ids(probs, 10, 0.99, 40, 0), 5, 1)
    dt = enc + beam
    return {"config": "cfg4 Efficient-Conformer B=64 10s beam=10", "encoder_ms": round(enc * 1e3, 2),
            "beam_ms": round(beam * 1e3, 2), "audio_s_per_s": round(640 / dt, 1)}


def cfg5():
    from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
    sd = squeezeformer_state_dict(vocab_size=V, seed=1234)
    conf = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=12, reduce_idx=5, recover_idx=11,
                feed_forward_expansion_factor=8, cnn_module_kernel=31)
    m = SqueezeformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd)
    rng = np.random.Generator(np.random.PCG64(20740))
    lens = np.sort(rng.integers(200, 3001, size=16))[::-1]
    # length buckets of width 200 frames, padded to the bucket maximum
    buckets = {}
    for l in lens:
        buckets.setdefault(int(l) // 200, []).append(int(l))
    batches = []
    for k in sorted(buckets, reverse=True):
        ls = buckets[k]
        x, l = synth_features(len(ls), max(ls), lens=ls, seed=20740 + k)
        batches.append((torch.from_numpy(x).cuda(), torch.from_numpy(l).cuda()))

    # a bucket of 1-3 utterances fills a fraction of the chip (24-70 row blocks, one beam workgroup per utterance):
    # every bucket runs on its own HIP stream so that buckets overlap
    streams = [torch.cuda.Stream() for _ in batches]

    def step():
        main = torch.cuda.current_stream()
        for (x, l), st in zip(batches, streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                probs = m.get_encoder_out(x, l)
                beam_search_ids(probs, 10, 0.99, 40, 0, frame_lens=torch.clamp((l + 3) // 4, max=probs.shape[1]).int())
        for st in streams:
            main.wait_stream(st)
    dt = timeit(step, 3, 1)
    out = [{"config": "cfg5 Squeezeformer 16 var-len utterances (2-30 s) in 200-frame buckets, beam=10",
            "buckets": len(batches), "ms": round(dt * 1e3, 2), "audio_s_per_s": round(float(lens.sum()) * 0.01 / dt, 1)}]
    # the same 16 utterances as ONE batch padded to the longest, without / with skip_padding (ragged-batch mode:
    # rows behind each utterance's valid frames are not computed)
    x, l = synth_features(16, int(lens[0]), lens=[int(v) for v in lens], seed=20740)
    x, l = torch.from_numpy(x).cuda(), torch.from_numpy(l).cuda()
    fl = torch.clamp((l + 3) // 4, max=m.out_frames(x.shape[1])).int()

    def one():
        probs = m.get_encoder_out(x, l)
        beam_search_ids(probs, 10, 0.99, 40, 0, frame_lens=fl)
    for skip in (False, True):
        m.set_skip_padding(skip)
        enc = timeit(lambda: m.get_encoder_out(x, l), 3, 1)
        dt = timeit(one, 3, 1)
        out.append({"config": "cfg5 as one padded batch of 16" + (", skip_padding" if skip else ""),
                    "encoder_ms": round(enc * 1e3, 2), "ms": round(dt * 1e3, 2),
                    "audio_s_per_s": round(float(lens.sum()) * 0.01 / dt, 1)})
    m.set_skip_padding(False)
    return out


if __name__ == "__main__":
    sel = sys.argv[1:] or ["cfg1", "cfg4", "cfg5"]
    for f in [globals()[n] for n in sel]:
        r = f()
        for line in (r if isinstance(r, list) else [r]):
            print(json.dumps(line), flush=True)
