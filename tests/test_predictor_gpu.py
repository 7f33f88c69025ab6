"""GPU tests of the drop-in surface B1/B2: PPASRPredictor.predict / predict_stream / reset_stream and
InferencePredictor (ppasr/predict.py, ppasr/infer_utils/inference_predictor.py) on synthetic audio."""
import numpy as np
import pytest
import torch

from ppasr_amd.utils.synth import conformer_state_dict, deepspeech2_state_dict, synth_vocabulary

pytestmark = pytest.mark.gpu


def _cfg(use_model="conformer", decoder="ctc_greedy", L=2):
    enc = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
    if use_model == "deepspeech2":
        enc = dict(num_rnn_layers=L, rnn_size=1024, use_gru=False)
    return dict(encoder_conf=enc, preprocess_conf=dict(feature_method="fbank", n_mels=80, sample_rate=16000,
                                                        use_dB_normalization=True, target_dB=-20),
                ctc_beam_search_decoder_conf=dict(alpha=2.2, beta=4.3, beam_size=10, num_processes=10, cutoff_prob=0.99,
                                                  cutoff_top_n=40, language_model_path=None),
                use_model=use_model, streaming=True, decoder=decoder, metrics_type="cer")


def _audio(seconds, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(int(16000 * seconds)) / 16000.0
    x = 0.1 * np.sin(2 * np.pi * 220 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.02 * rng.standard_normal(t.shape)
    return x.astype(np.float32)


@pytest.mark.parametrize("decoder", ["ctc_greedy", "ctc_beam_search"])
def test_predict_and_predict_stream_conformer(decoder):
    from ppasr_amd.predict import PPASRPredictor
    V = 300
    vocab = synth_vocabulary(V)
    sd = conformer_state_dict(vocab_size=V, num_blocks=2, seed=3)
    p = PPASRPredictor(configs=_cfg(decoder=decoder), state_dict=sd, vocab_list=vocab, warmup=True)
    wav = _audio(3.0)
    res = p.predict(audio_data=wav)
    assert set(res) == {"text", "score"} and isinstance(res["text"], str)
    # streaming: 0.5 s PCM16 chunks (infer_path.py:49-65 drives predict_stream this way)
    pcm = (np.clip(wav, -1, 1) * 32767).astype(np.int16).tobytes()
    step = 16000 * 2 // 2
    out = None
    n_none = 0
    for i in range(0, len(pcm), step):
        r = p.predict_stream(audio_data=pcm[i:i + step], is_end=(i + step >= len(pcm)))
        if r is None:
            n_none += 1
        else:
            out = r
    assert out is not None and isinstance(out["text"], str)
    assert n_none >= 1  # nothing is returned until 67 feature frames are buffered (predict.py:287)
    assert p.predictor.offset[0] > 0 and p.predictor.att_cache.shape[2] == p.predictor.offset[0]
    p.reset_stream()
    assert p.predictor.offset[0] == 0 and p.predictor.att_cache.shape == (0, 0, 0, 0)


def test_inference_predictor_matches_model_and_numpy_io():
    from ppasr_amd.infer_utils.inference_predictor import InferencePredictor
    V = 200
    sd = conformer_state_dict(vocab_size=V, num_blocks=2, seed=5)
    ip = InferencePredictor(_cfg(), "conformer", streaming=True, state_dict=sd)
    x = np.random.default_rng(0).standard_normal((1, 131, 80)).astype(np.float32) * 3 + 10
    probs = ip.predict(x, np.array([131], np.int64))
    assert isinstance(probs, np.ndarray) and probs.shape == (1, 32, V)
    c1 = ip.predict_chunk_conformer(x[:, :67], -16)
    c2 = ip.predict_chunk_conformer(x[:, 64:131], -16)
    assert c1.shape == (1, 16, V) and c2.shape == (1, 16, V) and int(ip.offset[0]) == 32
    assert ip.cnn_cache.shape == (2, 1, 256, 14) and ip.att_cache.shape == (2, 4, 32, 128)
    with pytest.raises(Exception):
        ip.predict_chunk_deepspeech(x)


def test_deepspeech2_predictor_streaming():
    from ppasr_amd.predict import PPASRPredictor
    V = 120
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=2, streaming=True, seed=7)
    p = PPASRPredictor(configs=_cfg("deepspeech2"), state_dict=sd, vocab_list=synth_vocabulary(V), warmup=False)
    wav = _audio(2.0, 1)
    assert isinstance(p.predict(audio_data=wav)["text"], str)
    pcm = (np.clip(wav, -1, 1) * 32767).astype(np.int16).tobytes()
    r = None
    for i in range(0, len(pcm), 32000):
        r = p.predict_stream(audio_data=pcm[i:i + 32000], is_end=(i + 32000 >= len(pcm))) or r
    assert r is not None
    assert p.predictor.output_state_h.shape == (2, 1, 1024)
    p.reset_stream()
    assert p.predictor.output_state_h is None


@pytest.mark.parametrize("use_model", ["squeezeformer", "efficient_conformer"])
def test_predict_stream_other_conformer_families(use_model):
    """predict_stream drives forward_chunk of every *former family (predict.py:300-309 -> predict_chunk_conformer)."""
    from ppasr_amd.predict import PPASRPredictor
    from ppasr_amd.utils.synth import efficient_conformer_state_dict, squeezeformer_state_dict
    V = 300
    vocab = synth_vocabulary(V)
    cfg = _cfg(use_model=use_model, L=4)
    if use_model == "squeezeformer":
        cfg["encoder_conf"] = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=4, reduce_idx=1,
                                   recover_idx=3, feed_forward_expansion_factor=8, cnn_module_kernel=31)
        sd = squeezeformer_state_dict(vocab_size=V, num_blocks=4, seed=5)
    else:
        cfg["encoder_conf"] = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=4, cnn_module_kernel=15,
                                   cnn_module_norm="layer_norm",
                                   efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3,
                                                       stride_kernel=True))
        sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=4, seed=5, stride_layer_idx=1, group_layer_idx=(0, 1))
    p = PPASRPredictor(configs=cfg, state_dict=sd, vocab_list=vocab, warmup=False)
    wav = _audio(2.6)
    pcm = (np.clip(wav, -1, 1) * 32767).astype(np.int16).tobytes()
    step = 16000 * 2 // 2
    out = None
    for i in range(0, len(pcm), step):
        r = p.predict_stream(audio_data=pcm[i:i + step], is_end=False)
        out = r or out
    assert out is not None and isinstance(out["text"], str)
    per_chunk = 8 if use_model == "efficient_conformer" else 16
    assert p.predictor.offset[0] > 0 and p.predictor.offset[0] % per_chunk == 0
    assert p.predictor.att_cache.shape[2] == p.predictor.offset[0] * (16 // per_chunk)
    p.reset_stream()
    assert p.predictor.offset[0] == 0


def test_evaluate_harness_greedy_cer():
    """trainer.evaluate's decode half: labels = the model's own greedy output -> CER 0; perturbed labels -> CER > 0."""
    from ppasr_amd.evaluate import evaluate
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    from ppasr_amd.utils.synth import synth_features
    V = 120
    vocab = synth_vocabulary(V)
    sd = conformer_state_dict(vocab_size=V, num_blocks=2, seed=9)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=2, cnn_module_kernel=15)
    model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    batches = []
    for seed in (1, 2):
        x, lens = synth_features(3, 203, lens=[203, 203, 203], seed=seed)
        tokens, n, _ = model.encode_greedy(x, lens)
        labels = tokens.cpu().numpy().copy()
        labels[labels == V - 1] = 2  # <eos> is stripped from labels (utils.py:62); keep the strings comparable
        labels[labels == 1] = 2      # so is <unk>
        batches.append((x, labels, lens, n.cpu().numpy()))
    # predictions still contain the original ids, so rebuild the expectation from the label strings themselves
    from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decoder_batch
    from ppasr_amd.utils.metrics import cer, labels_to_string
    want, cnt = 0.0, 0
    for x, labels, lens, _ in batches:
        outs = greedy_decoder_batch(model.get_encoder_out(x, lens), vocab)
        for o, l in zip(outs, labels_to_string(labels, vocab, eos=V - 1)):
            want += cer(o, l)
            cnt += 1
    got = evaluate(model, batches, vocab, decoder="ctc_greedy", metrics_type="cer")
    assert got == pytest.approx(want / cnt) and 0.0 <= got < 0.2


def test_stream_pool_equals_predict_stream_per_session():
    """serving.StreamPool (session groups) reproduces PPASRPredictor.predict_stream for every session."""
    from ppasr_amd.predict import PPASRPredictor
    from ppasr_amd.serving import StreamPool
    V = 300
    vocab = synth_vocabulary(V)
    sd = conformer_state_dict(vocab_size=V, num_blocks=2, seed=3)
    cfg = _cfg(decoder="ctc_greedy")
    wavs = [_audio(2.4, seed=s) for s in range(3)]
    pcms = [(np.clip(w, -1, 1) * 32767).astype(np.int16).tobytes() for w in wavs]
    step = 16000 * 2 // 2  # 0.5 s packets
    want = []
    p = PPASRPredictor(configs=cfg, state_dict=sd, vocab_list=vocab, warmup=False)
    for pcm in pcms:
        p.reset_stream()
        out = None
        for i in range(0, len(pcm), step):
            out = p.predict_stream(audio_data=pcm[i:i + step], is_end=False) or out
        want.append(out)
    pool = StreamPool(p.predictor.model, vocab, n_sessions=3, preprocess_conf=cfg["preprocess_conf"])
    got = [None] * 3
    for i in range(0, max(len(x) for x in pcms), step):
        for s, pcm in enumerate(pcms):
            if i < len(pcm) and not (s == 2 and i == 0):  # session 2 joins one packet late...
                pool.feed(s, pcm[i - step if s == 2 else i:][:step])
        for s, r in pool.step().items():
            got[s] = r
    pool.feed(2, pcms[2][len(pcms[2]) - step:])  # ...and receives its last packet afterwards
    for s, r in pool.step().items():
        got[s] = r
    for s in range(3):
        assert got[s] is not None and want[s] is not None
        assert got[s]["text"] == want[s]["text"], s
        assert abs(got[s]["score"] - want[s]["score"]) < 1e-3


@pytest.mark.parametrize("decoder", ["ctc_greedy", "ctc_beam_search"])
def test_evaluate_overlapped_decode_gives_the_same_result(decoder):
    """evaluate(overlap_decode=True) encodes batch i+1 on a second HIP stream while batch i is decoded: same error
    rate as the serial loop, with ragged batches and trimming included."""
    from ppasr_amd.decoders.beam_search_decoder import BeamSearchDecoder
    from ppasr_amd.evaluate import evaluate
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    from ppasr_amd.utils.synth import synth_features
    V = 120
    vocab = synth_vocabulary(V)
    sd = conformer_state_dict(vocab_size=V, num_blocks=2, seed=19)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=2, cnn_module_kernel=15)
    model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    bsd = BeamSearchDecoder(0.0, 0.0, 20, 0.99, 40, vocab) if decoder == "ctc_beam_search" else None
    rng = np.random.Generator(np.random.PCG64(3))
    batches = []
    for seed in range(5):
        lens = sorted((int(v) for v in rng.integers(60, 400, size=4)), reverse=True)
        x, la = synth_features(4, lens[0], lens=lens, seed=seed)
        labels = rng.integers(2, V - 1, size=(4, 25)).astype(np.int64)
        batches.append((x, labels, la, None))
    for trim in (False, True):
        a = evaluate(model, batches, vocab, decoder=decoder, beam_search_decoder=bsd, trim_padding=trim, overlap_decode=False)
        b = evaluate(model, batches, vocab, decoder=decoder, beam_search_decoder=bsd, trim_padding=trim, overlap_decode=True)
        assert a == b and a >= 0


@pytest.mark.timeout(300)
@pytest.mark.parametrize("beam", [20, 48])
def test_beam_search_beside_the_encoder_is_deterministic(beam):
    """Round 6: the search of step i shares CUs with the encoder of step i + 1 (evaluate(overlap_decode=True)); waves of
    its workgroup are then delayed unevenly, and a word of the selection that thread 0 re-armed in the barrier-less tail of
    a frame was read late by another wave (wrong survivor count -> a different beam, or a hang).  The selection words now
    alternate with the frame's parity (ctc_beam.hip); this is the stress that showed 5-7 differing decodes in 300."""
    from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
    from ppasr_amd.model_utils.conformer.model import ConformerModel
    from ppasr_amd.utils.synth import synth_features
    V = 120
    sd = conformer_state_dict(vocab_size=V, num_blocks=2, seed=19)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=2, cnn_module_kernel=15)
    model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
    rng = np.random.Generator(np.random.PCG64(3))
    batches = []
    for seed in range(5):
        lens = sorted((int(v) for v in rng.integers(60, 400, size=4)), reverse=True)
        batches.append(synth_features(4, lens[0], lens=lens, seed=seed))
    tables = [model.get_encoder_out(x, la).clone() for x, la in batches]
    want = []
    for q in tables:
        t, n, s, _ = beam_search_ids(q, beam, 0.99, 40, 0, nbest=1)
        torch.cuda.synchronize()
        want.append((t.clone(), n.clone(), s.clone()))
    side = torch.cuda.Stream()
    for _ in range(30):
        for i, q in enumerate(tables):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model.get_encoder_out(*batches[(i + 1) % 5])
            t, n, s, _ = beam_search_ids(q, beam, 0.99, 40, 0, nbest=1)
            torch.cuda.synchronize()
            assert torch.equal(t, want[i][0]) and torch.equal(n, want[i][1]) and torch.equal(s, want[i][2])


def test_stream_pool_finish_equals_predict_stream_is_end():
    """StreamPool.finish flushes the last, shorter window like predict_stream(is_end=True) (predict.py:291-298)."""
    from ppasr_amd.predict import PPASRPredictor
    from ppasr_amd.serving import StreamPool
    V = 300
    vocab = synth_vocabulary(V)
    sd = conformer_state_dict(vocab_size=V, num_blocks=2, seed=3)
    cfg = _cfg(decoder="ctc_greedy")
    wavs = [_audio(1.93, seed=11), _audio(2.71, seed=12)]
    pcms = [(np.clip(w, -1, 1) * 32767).astype(np.int16).tobytes() for w in wavs]
    step = 16000  # 0.5 s packets
    p = PPASRPredictor(configs=cfg, state_dict=sd, vocab_list=vocab, warmup=False)
    want = []
    for pcm in pcms:
        p.reset_stream()
        out = None
        for i in range(0, len(pcm), step):
            out = p.predict_stream(audio_data=pcm[i:i + step], is_end=(i + step >= len(pcm))) or out
        want.append(out)
    pool = StreamPool(p.predictor.model, vocab, n_sessions=2, preprocess_conf=cfg["preprocess_conf"])
    for i in range(0, max(len(x) for x in pcms), step):
        for s, pcm in enumerate(pcms):
            if i < len(pcm):
                pool.feed(s, pcm[i:i + step])
        pool.step()
    for s in range(2):
        got = pool.finish(s)
        assert got is not None and got["text"] == want[s]["text"], s
        assert abs(got["score"] - want[s]["score"]) < 1e-3


def test_predict_long_with_given_segments():
    """predict.py:190-229 with the VAD result supplied by the caller: per-segment predict, texts joined with '，',
    mean score."""
    from ppasr_amd.predict import PPASRPredictor
    V = 300
    vocab = synth_vocabulary(V)
    sd = conformer_state_dict(vocab_size=V, num_blocks=2, seed=3)
    p = PPASRPredictor(configs=_cfg(decoder="ctc_greedy"), state_dict=sd, vocab_list=vocab, warmup=False)
    wav = _audio(6.0, seed=4)
    segs = [{"start": 1600, "end": 30000}, {"start": 40000, "end": 70000}, {"start": 72000, "end": 96000}]
    parts = [p.predict(audio_data=wav[t["start"]:t["end"]]) for t in segs]
    res = p.predict_long(audio_data=wav, speech_timestamps=segs)
    assert res["text"] == "，".join(r["text"] for r in parts if r["text"] != "")
    assert res["score"] == round(sum(r["score"] for r in parts) / len(parts), 2)

    class _Vad:
        def get_speech_timestamps(self, samples, sr):
            assert sr == 16000 and len(samples) == len(wav)
            return segs[:2]
    assert p.predict_long(audio_data=wav, vad_predictor=_Vad())["text"] == "，".join(r["text"] for r in parts[:2] if r["text"])
    with pytest.raises(NotImplementedError):
        p.predict_long(audio_data=wav)


# ---- the reference's own audio file through the drop-in surface ------------------------------------------------------
def _wav_fixture():
    import os
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_wav.npz")) as z:
        return {k: z[k] for k in z.files}


def _collapse_text(ids, vocab):
    keep = np.concatenate([[True], ids[1:] != ids[:-1]]) & (ids != 0)
    return "".join(vocab[i] for i in ids[keep]).replace("<space>", " ")


def test_predict_reference_wav_gpu(tmp_path):
    """/root/reference/dataset/test.wav (the file docs/infer.md:93 recognises; carried in tests/golden/ref_wav.npz as int16
    samples) through PPASRPredictor.predict (file path, predict.py:163-187) and through predict_stream in 0.5 s PCM chunks
    (predict.py:232-337), against what the REFERENCE's own PPASRPredictor source returned for the same file and the same
    random-init configs/conformer.yml model (tests/golden/make_wav_goldens.py: reference predict.py / audio.py /
    audio_featurizer.py / inference_predictor.py / ctc_greedy_decoder.py / conformer model source, on the shim, with
    oracle/fbank_oracle.py as paddleaudio's fbank).  Token for token; every None / text of the streaming session."""
    import os
    import wave

    import yaml
    from ppasr_amd.predict import PPASRPredictor
    from ppasr_amd.utils.synth import synth_vocabulary
    z = _wav_fixture()
    V = int(z["vocab_size"])
    vocab = synth_vocabulary(V)
    sd = conformer_state_dict(vocab_size=V, num_blocks=12, seed=int(z["sd_seed"]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "configs", "conformer.yml"), "r", encoding="utf-8") as f:
        cfg = yaml.load(f.read(), Loader=yaml.FullLoader)
    cfg["decoder"] = "ctc_greedy"
    p = PPASRPredictor(configs=cfg, state_dict=sd, vocab_list=vocab, warmup=True)
    path = str(tmp_path / "test.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(z["sample_rate"]))
        w.writeframes(z["samples"].tobytes())

    # (1) the feature front end on this audio: the reference-side features (float64 Kaldi restatement, every 16th frame)
    feats = p._audio_featurizer.featurize(z["samples"].astype(np.float32) / 32768.0, int(z["sample_rate"]))
    assert feats.shape == (int(z["n_feature_frames"]), 80)
    d_f = np.abs(feats[::16] - z["feats_16"])
    e_f = float(d_f.max())
    print(f"test.wav: {feats.shape[0]} feature frames, |fbank - reference-side fbank| max {e_f:.2e}, "
          f"{int((d_f > 1e-3).sum())}/{d_f.size} values beyond 1e-3")
    # The gain of AudioSegment.normalize goes through np.log10 on a float32 scalar: numpy's float32 log10 on the machine
    # that made the fixture is 1 ulp off the correctly rounded value the device computes, the gain differs in its last
    # bits, and 1 in 2000 int16 samples (clip + TRUNCATE, audio.py:549-577) lands one LSB away -- log-mel values of a few
    # low-energy bins move by up to 5e-3.  The reference's own result has that platform dependence; everything else
    # agrees to 1e-4.
    assert e_f < 1e-2 and (d_f > 1e-3).mean() < 0.01

    # (2) whole utterance: frame ids, then the text and the score of predict()
    probs = p.predictor.predict_device(feats[np.newaxis], np.array([feats.shape[0]], np.int64))[0].cpu().numpy()
    assert probs.shape[0] == z["ids"].shape[0]
    ids = probs.argmax(-1)
    differ = np.nonzero(ids != z["ids"])[0]
    print(f"predict: {len(ids)} frames, ids differing from the reference's: {len(differ)} "
          f"(reference top-2 probability margins there: {z['margin'][differ]}; smallest margin of the utterance {z['margin'].min():.1e})")
    assert np.all(z["margin"][differ] < 1e-4), differ  # only a near-tie of the reference itself may differ
    assert float(np.abs(probs.max(-1) - z["maxprob"]).max()) < 1e-3
    res = p.predict(audio_data=path)
    assert set(res) == {"text", "score"}
    assert res["text"] == _collapse_text(ids, vocab)            # predict() == the collapse of the HIP path's own frames
    if len(differ) == 0:
        assert res["text"] == str(z["predict_text"])              # == the reference's PPASRPredictor.predict text
        assert abs(res["score"] - float(z["predict_score"])) < 5e-2  # (score = 100 x mean max-prob of non-blank frames)

    # (3) streaming session: 0.5 s PCM16 chunks, is_end on the last (infer_path.py:49-65)
    pcm = z["samples"].tobytes()
    step = int(int(z["sample_rate"]) * float(z["chunk_seconds"])) * 2
    p.reset_stream()
    calls = range(0, len(pcm), step)
    assert len(calls) == len(z["stream_none"])
    n_text_equal = n_text = 0
    for k, i in enumerate(calls):
        r = p.predict_stream(audio_data=pcm[i:i + step], is_end=(i + step >= len(pcm)))
        assert (r is None) == bool(z["stream_none"][k]), k
        if r is not None:
            n_text += 1
            n_text_equal += int(r["text"] == str(z["stream_texts"][k]))
    assert int(p.predictor.offset[0]) == int(z["stream_out_frames"])
    final = r
    print(f"predict_stream: {len(calls)} calls, {n_text} texts, {n_text_equal} equal to the reference's; smallest reference "
          f"top-2 margin of the session {z['stream_margin'].min():.1e}")
    if z["stream_margin"].min() > 1e-4:
        assert n_text_equal == n_text
        assert final["text"] == str(z["stream_texts"][-1])
        assert abs(final["score"] - float(z["stream_scores"][-1])) < 5e-2
    p.reset_stream()


@pytest.mark.parametrize("use_model", ["squeezeformer", "efficient_conformer"])
def test_stream_pool_serves_the_other_former_families(use_model):
    """serving.StreamPool on a Squeezeformer / Efficient-Conformer model: the library has no session-group call for these
    handles, so the pool drives one stream handle per session behind the group interface (StreamHandleSet) -- every session
    still reproduces its own PPASRPredictor.predict_stream, the late joiner and the final `finish` included."""
    from ppasr_amd.model_utils.conformer.model import StreamHandleSet
    from ppasr_amd.predict import PPASRPredictor
    from ppasr_amd.serving import StreamPool
    from ppasr_amd.utils.synth import efficient_conformer_state_dict, squeezeformer_state_dict
    V = 300
    vocab = synth_vocabulary(V)
    cfg = _cfg(use_model=use_model, L=4)
    if use_model == "squeezeformer":
        cfg["encoder_conf"] = dict(encoder_dim=256, output_size=256, attention_heads=4, num_blocks=4, reduce_idx=1,
                                   recover_idx=3, feed_forward_expansion_factor=8, cnn_module_kernel=31)
        sd = squeezeformer_state_dict(vocab_size=V, num_blocks=4, seed=5)
    else:
        cfg["encoder_conf"] = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=4, cnn_module_kernel=15,
                                   cnn_module_norm="layer_norm",
                                   efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3,
                                                       stride_kernel=True))
        sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=4, seed=5, stride_layer_idx=1, group_layer_idx=(0, 1))
    p = PPASRPredictor(configs=cfg, state_dict=sd, vocab_list=vocab, warmup=False)
    wavs = [_audio(2.4, seed=21), _audio(1.93, seed=22)]
    pcms = [(np.clip(w, -1, 1) * 32767).astype(np.int16).tobytes() for w in wavs]
    step = 16000  # 0.5 s packets
    want = []
    for pcm in pcms:
        p.reset_stream()
        out = None
        for i in range(0, len(pcm), step):
            out = p.predict_stream(audio_data=pcm[i:i + step], is_end=(i + step >= len(pcm))) or out
        want.append(out)
    p.reset_stream()
    pool = StreamPool(p.predictor.model, vocab, n_sessions=2, preprocess_conf=cfg["preprocess_conf"])
    assert isinstance(pool.group, StreamHandleSet)
    for i in range(0, max(len(x) for x in pcms), step):
        for s, pcm in enumerate(pcms):
            if i < len(pcm):
                pool.feed(s, pcm[i:i + step])
        pool.step()
    for s in range(2):
        got = pool.finish(s)
        assert got is not None and want[s] is not None and got["text"] == want[s]["text"], s
        assert abs(got["score"] - want[s]["score"]) < 1e-3
    pool.reset(0)
    assert pool.group.offset(0) == 0
