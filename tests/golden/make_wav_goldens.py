"""Regenerate tests/golden/ref_wav.npz: the reference's OWN `PPASRPredictor` (ppasr/predict.py, unmodified source) run on
the reference's own audio file (/root/reference/dataset/test.wav, the file docs/infer.md:93 recognises).

    python tests/golden/make_wav_goldens.py

How.  `sys.path[:0] = [oracle/paddle_shim, /root/reference]`; the reference's `PPASRPredictor.__init__`, `predict`,
`predict_stream`, `reset_stream`, `AudioSegment`, `AudioFeaturizer`, `InferencePredictor`, `greedy_decoder` and
`greedy_decoder_chunk` execute from source.  What the shim supplies underneath them:
  * `paddle.inference` (oracle/paddle_shim/paddle/inference.py): handles + `run()` over the function the reference's
    `ConformerModel.export()` returns (`get_encoder_out_chunk`, conformer/model.py:187-206), i.e. the reference's model
    source in dygraph instead of the exported ProgramDesc;
  * `paddleaudio.compliance.kaldi.fbank` = oracle/fbank_oracle.py (paddleaudio is not in /root/reference: PARITY UNPINNED
    for the feature arithmetic -- this fixture pins everything AROUND it: audio loading, dB normalisation, int16 scaling,
    the whole-utterance call through the chunk graph, the 67 / 64-frame window state machine, cache / offset carry, the
    greedy collapse and its stateful chunk form);
  * `soundfile.read` through the standard library's `wave`.
Weights: random-init `configs/conformer.yml` model (12 blocks, V = 4233, the seeds below); there is no checkpoint offline.

Stored: the audio itself (int16 samples -- the file cannot travel, /root/reference does not exist on the GPU box), the
`predict` result, every `predict_stream` result of a 0.5 s-chunk session (infer_path.py:49-65 drives it that way), the
reference's per-frame ids / top-2 probability margins of the whole-utterance pass (so a test can tell a near-tie frame
from a wrong one), and the reference's fbank output sub-sampled (every 16th frame) for diagnosis.
"""
import json
import os
import sys
import tempfile
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
WAV = os.path.join(REFERENCE, "dataset", "test.wav")
sys.path.insert(0, ROOT)

V, SD_SEED, CHUNK_SECONDS = 4233, int(os.environ.get("WAV_SD_SEED", 4322)), 0.5


def main():
    if not os.path.isdir(REFERENCE):
        raise SystemExit("needs /root/reference (build container only)")
    from ppasr_amd.utils.synth import conformer_state_dict, synth_vocabulary
    sys.path[:0] = [os.path.join(ROOT, "oracle", "paddle_shim"), REFERENCE]
    import paddle
    import paddle.inference as paddle_infer
    import torch
    import yaml
    torch.set_grad_enabled(False)
    if not hasattr(np, "sctypes"):  # numpy 1.x table the reference reads (data_utils/audio.py:533-560); removed in numpy 2
        np.sctypes = {"int": [np.int8, np.int16, np.int32, np.int64], "uint": [np.uint8, np.uint16, np.uint32, np.uint64],
                      "float": [np.float16, np.float32, np.float64, np.longdouble],
                      "complex": [np.complex64, np.complex128, np.clongdouble], "others": [bool, object, bytes, str, np.void]}
    from ppasr.data_utils.audio import AudioSegment
    from ppasr.model_utils.conformer.model import ConformerModel
    from ppasr.predict import PPASRPredictor

    # numpy 1.x scalar promotion in AudioSegment.rms_db / normalize / gain_db (data_utils/audio.py:519-530,287-304,256-264).
    # The reference only runs on numpy 1.x (np.sctypes above).  There an operation between a numpy scalar and a Python
    # number promotes like two arrays (legacy rule for all-scalar operands; NEP 50's table: `uint8(1) + 2 -> int64`,
    # `float32(1) + 3e100 -> float64`): `10 * np.log10(mean_square)` is int64 x float32 = float64 -- ten times the float32
    # logarithm, exactly --, `target_db - rms_db` and `gain / 20.` stay float64, the power is taken in float64 and rounded
    # to float32 once, when it scales the float32 samples.  numpy 2 (NEP 50) keeps rms_db and the gain in float32 instead
    # (3 ulp of the linear gain; about 1 int16 sample in 2 000 one LSB away, log-mel values up to 5e-3 away).  Handing the
    # two methods Python floats reproduces the numpy 1.x arithmetic on numpy 2: rms_db below is the reference's property
    # with the float32 logarithm converted before the multiplication.
    def _rms_db_numpy1(self):
        mean_square = np.mean(self._samples ** 2)  # float32: pairwise sums of 8192-element chunks, both numpy versions
        if mean_square == 0:
            mean_square = 1
        return 10 * float(np.log10(mean_square))

    AudioSegment.rms_db = property(_rms_db_numpy1)
    _gain_db = AudioSegment.gain_db
    AudioSegment.gain_db = lambda self, gain: _gain_db(self, float(gain))

    with open(os.path.join(REFERENCE, "configs", "conformer.yml"), "r", encoding="utf-8") as f:
        configs = yaml.load(f.read(), Loader=yaml.FullLoader)
    assert configs["use_model"] == "conformer" and configs["streaming"]
    configs["decoder"] = "ctc_greedy"  # (the YAML's ctc_beam_search needs paddlespeech_ctcdecoders, absent: predict.py:93-105)
    tmp = tempfile.mkdtemp()
    vocab = synth_vocabulary(V)
    vocab_path = os.path.join(tmp, "vocabulary.txt")
    with open(vocab_path, "w", encoding="utf-8") as f:
        for i, tok in enumerate(vocab):
            f.write(f"{tok}\t{V - i}\n")
    configs["dataset_conf"]["dataset_vocab"] = vocab_path

    # the model, built the way trainer.py:172-210 does and loaded by name; "exported" = model.export()
    sd = conformer_state_dict(vocab_size=V, num_blocks=configs["encoder_conf"]["num_blocks"], seed=SD_SEED)
    mean_istd = os.path.join(tmp, "mean_istd.json")
    with open(mean_istd, "w") as f:
        json.dump({"mean": sd["encoder.global_cmvn.mean"].tolist(), "istd": sd["encoder.global_cmvn.istd"].tolist()}, f)
    model = ConformerModel(input_dim=80, vocab_size=V, mean_istd_path=mean_istd, streaming=True,
                           encoder_conf=configs["encoder_conf"], decoder_conf=configs["decoder_conf"],
                           **configs["model_conf"])
    missing, unexpected = model.set_state_dict(sd)
    assert not unexpected and all(m.startswith("decoder.") for m in missing), (missing[:5], unexpected[:5])
    model.eval()
    model_dir = os.path.join(tmp, "infer")
    os.makedirs(model_dir)
    for n in ("model.pdmodel", "model.pdiparams"):  # InferencePredictor only checks that they exist (:41-45)
        open(os.path.join(model_dir, n), "wb").close()
    paddle_infer.register(model_dir, model.export(), ["speech", "offset", "required_cache_size", "att_cache", "cnn_cache"], 3)

    captured = {}
    orig_run = paddle_infer._Predictor.run

    def run_and_capture(self):  # keep the whole-utterance probabilities of the `predict` call for the margins
        r = orig_run(self)
        captured["probs"] = self._out["output_0"].value
        captured["speech"] = self._in["speech"].value
        captured.setdefault("all", []).append(captured["probs"][0])
        captured.setdefault("inputs", []).append(captured["speech"][0].copy())
        return r

    paddle_infer._Predictor.run = run_and_capture
    np.random.seed(0)  # (the constructor's warm-up draws its audio from numpy's global generator)
    p = PPASRPredictor(configs=configs, model_path=model_dir, use_gpu=False)

    res = p.predict(audio_data=WAV)
    probs, feats = captured["probs"][0], captured["speech"][0]
    ids = probs.argmax(-1).astype(np.int32)
    top2 = np.sort(probs, axis=-1)[:, -2:]
    margin = (top2[:, 1] - top2[:, 0]).astype(np.float32)
    print(f"[ref] predict: {feats.shape[0]} feature frames -> {probs.shape[0]} output frames; "
          f"{len(res['text'])} characters, score {res['score']:.4f}; min top-2 margin {margin.min():.2e}")

    with wave.open(WAV, "rb") as w:
        sr, ch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        pcm = w.readframes(n)
    assert (sr, ch, sw) == (16000, 1, 2)
    step = int(sr * CHUNK_SECONDS) * sw
    stream_texts, stream_scores, stream_none = [], [], []
    p.reset_stream()
    captured["all"], captured["inputs"] = [], []
    for i in range(0, len(pcm), step):
        r = p.predict_stream(audio_data=pcm[i:i + step], is_end=(i + step >= len(pcm)))
        stream_none.append(r is None)
        stream_texts.append("" if r is None else r["text"])
        stream_scores.append(np.nan if r is None or r["score"] is None else float(r["score"]))
    n_out = int(p.predictor.offset[0])
    p.reset_stream()
    sp = np.concatenate(captured["all"], 0)
    assert sp.shape[0] == n_out
    s_top2 = np.sort(sp, axis=-1)[:, -2:]
    s_ids, s_margin = sp.argmax(-1).astype(np.int32), (s_top2[:, 1] - s_top2[:, 0]).astype(np.float32)
    print(f"[ref] predict_stream: min top-2 margin {s_margin.min():.2e}")
    # the feature windows the reference's state machine fed to the model (67 frames, the last one shorter): their sizes
    # and float64 sums -- they pin the per-call featurisation, the IN-PLACE dB normalisation of the buffered samples
    # (audio_featurizer.py:48-50 on predict.py:262-268's `remained_wav`) and the window / stride / cache arithmetic
    win_frames = np.array([w.shape[0] for w in captured["inputs"]], np.int32)
    win_sums = np.array([w.astype(np.float64).sum() for w in captured["inputs"]], np.float64)
    win_first = np.stack([w[0] for w in captured["inputs"]]).astype(np.float32)
    print(f"[ref] predict_stream: {len(stream_none)} calls, {sum(stream_none)} returned None, {n_out} output frames, "
          f"final text {len(stream_texts[-1])} characters (== predict's: {stream_texts[-1] == res['text']})")

    out = dict(samples=np.frombuffer(pcm, np.int16).copy(), sample_rate=np.int32(sr),
               predict_text=np.array(res["text"]), predict_score=np.float64(res["score"]),
               ids=ids, margin=margin, maxprob=top2[:, 1].astype(np.float32), feats_16=feats[::16].astype(np.float32),
               n_feature_frames=np.int32(feats.shape[0]),
               stream_texts=np.array(stream_texts), stream_scores=np.array(stream_scores, np.float64),
               stream_none=np.array(stream_none), stream_out_frames=np.int32(n_out), stream_ids=s_ids,
               stream_margin=s_margin, stream_win_frames=win_frames, stream_win_sums=win_sums, stream_win_first=win_first,
               chunk_seconds=np.float64(CHUNK_SECONDS), vocab_size=np.int32(V), sd_seed=np.int32(SD_SEED))
    np.savez_compressed(os.path.join(HERE, "ref_wav.npz"), **out)
    print("wrote ref_wav.npz", os.path.getsize(os.path.join(HERE, "ref_wav.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
