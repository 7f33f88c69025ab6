"""Drop-in for ``ppasr/predict.py`` (``PPASRPredictor``): ``predict`` (:163-187), ``predict_stream``
(:232-337), ``reset_stream`` (:340-347), ``decode`` (:114-140) with the same arguments and return
dicts.  Audio -> fbank is host-side glue (``data_utils/featurizer.py``); encoder + CTC decode run in
the HIP library.  VAD (`predict_long`), punctuation and ITN are separate models outside the hot
path and are not provided (``use_pun=True`` / ``is_itn=True`` raise).
"""
import numpy as np
import torch
import yaml

from ppasr_amd.data_utils.featurizer import AudioFeaturizer, TextFeaturizer, db_gain, load_audio, pcm_bytes_to_float
from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decoder, greedy_decoder_chunk
from ppasr_amd.infer_utils.inference_predictor import InferencePredictor

__all__ = ["PPASRPredictor"]


class _Cfg(dict):
    """dict_to_object (utils/utils.py:45-56): nested attribute access."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return _Cfg(v) if isinstance(v, dict) else v

    def __setattr__(self, k, v):
        self[k] = v


class PPASRPredictor:
    def __init__(self, configs=None, model_tag=None, model_path="models/conformer_streaming_fbank/infer/",
                 use_pun=False, pun_model_dir="models/pun_models/", use_gpu=True, state_dict=None, vocab_list=None,
                 warmup=True):
        if configs is None:
            raise Exception("configs (yaml path or dict) is required: model download is unavailable offline")
        if isinstance(configs, str):
            with open(configs, "r", encoding="utf-8") as f:
                configs = yaml.load(f.read(), Loader=yaml.FullLoader)
        if use_pun:
            raise NotImplementedError("punctuation model is outside the hot path (SURVEY.md §2 row 19)")
        self.configs = _Cfg(configs)
        self.running = False
        if vocab_list is not None:
            self._text_featurizer = TextFeaturizer(vocab_list=vocab_list)
        else:
            self._text_featurizer = TextFeaturizer(vocab_filepath=self.configs.dataset_conf.dataset_vocab)
        self._audio_featurizer = AudioFeaturizer(**self.configs.preprocess_conf)
        self.remained_wav = None
        self.cached_feat = None
        self.greedy_last_max_prob_list = None
        self.greedy_last_max_index_list = None
        self.beam_search_decoder = None
        if self.configs.decoder == "ctc_beam_search":
            from ppasr_amd.decoders.beam_search_decoder import BeamSearchDecoder
            self.beam_search_decoder = BeamSearchDecoder(vocab_list=self._text_featurizer.vocab_list,
                                                         **self.configs.ctc_beam_search_decoder_conf)
        self.predictor = InferencePredictor(configs=self.configs, use_model=self.configs.use_model,
                                            streaming=self.configs.streaming, model_dir=model_path, use_gpu=use_gpu,
                                            state_dict=state_dict, vocab_size=self._text_featurizer.vocab_size)
        if warmup:  # predict.py:88-89
            self.predict(audio_data=np.random.uniform(low=-2.0, high=2.0, size=(134240,)).astype(np.float32))

    def decode(self, output_data, use_pun=False, is_itn=False):
        """predict.py:114-140"""
        if use_pun or is_itn:
            raise NotImplementedError("punctuation / ITN are outside the hot path")
        if self.configs.decoder == "ctc_beam_search":
            result = self.beam_search_decoder.decode_beam_search_offline(probs_split=output_data)
        else:
            result = greedy_decoder(probs_seq=output_data, vocabulary=self._text_featurizer.vocab_list)
        return result[0], result[1]

    def predict(self, audio_data, use_pun=False, is_itn=False, sample_rate=16000):
        """-> {'text': str, 'score': float}.  predict.py:163-187"""
        samples, sr = load_audio(audio_data, sample_rate)
        feat = self._audio_featurizer.featurize(samples, sr)
        input_data = np.asarray(feat, np.float32)[np.newaxis, :]
        audio_len = np.array([input_data.shape[1]]).astype(np.int64)
        probs = self.predictor.predict_device(input_data, audio_len)[0]  # stays on the device
        score, text = self.decode(output_data=probs, use_pun=use_pun, is_itn=is_itn)
        return {"text": text, "score": score}

    def predict_long(self, audio_data, use_pun=False, is_itn=False, sample_rate=16000, speech_timestamps=None,
                     vad_predictor=None):
        """predict.py:190-229: recognise the speech segments of a long recording one by one and join the texts with
        '，'; score = mean of the segment scores (2 decimals).  The reference finds the segments with its Silero VAD
        model (``init_vad``), which is a separate model outside this path: pass the segments as ``speech_timestamps``
        (``[{'start': sample, 'end': sample}, ...]``, what ``get_speech_timestamps`` returns) or an object with that
        method as ``vad_predictor``."""
        if use_pun:
            raise NotImplementedError("punctuation is a separate model outside the hot path")
        samples, sr = load_audio(audio_data, sample_rate)
        if speech_timestamps is None:
            if vad_predictor is None:
                raise NotImplementedError("predict_long needs speech_timestamps or a vad_predictor: the Silero VAD "
                                          "model is not part of this library")
            speech_timestamps = vad_predictor.get_speech_timestamps(samples, sr)
        texts, scores = "", []
        for t in speech_timestamps:
            result = self.predict(audio_data=samples[t["start"]:t["end"]], use_pun=False, is_itn=is_itn, sample_rate=sr)
            score, text = result["score"], result["text"]
            if text != "":
                texts = texts + "，" + text
            scores.append(score)
        if texts[:1] == "，":
            texts = texts[1:]
        return {"text": texts, "score": round(sum(scores) / len(scores), 2) if scores else 0}

    def predict_stream(self, audio_data, is_end=False, use_pun=False, is_itn=False, channels=1, samp_width=2,
                       sample_rate=16000):
        """predict.py:232-337 (same windowing constants: 16 output frames per 67-frame window, stride 64,
        3 cached feature frames, required_cache_size = -16)."""
        if not self.configs.streaming:
            raise Exception(f"不支持改该模型流式识别，当前模型：{self.configs.use_model}")
        if isinstance(audio_data, np.ndarray):
            samples, _ = load_audio(audio_data, sample_rate)
        elif isinstance(audio_data, bytes):
            samples = pcm_bytes_to_float(audio_data, channels, samp_width)
        else:
            raise Exception(f"不支持该数据类型，当前数据类型为：{type(audio_data)}")
        self.remained_wav = samples if self.remained_wav is None else np.concatenate([self.remained_wav, samples])
        x_chunk = self._audio_featurizer.featurize(self.remained_wav, sample_rate)
        x_chunk = np.asarray(x_chunk, np.float32)[np.newaxis, :]
        self.cached_feat = x_chunk if self.cached_feat is None else np.concatenate([self.cached_feat, x_chunk], axis=1)
        # The reference's featurize() normalises the AudioSegment it is given IN PLACE (audio_featurizer.py:48-50 ->
        # audio.py:287-304, `self._samples *= gain`), and here that segment is the buffered `remained_wav`: the samples
        # that stay buffered for the next call are the GAINED ones, and they are gained again with the next chunk's
        # factor.  Reproduced (found by tests/golden/ref_wav.npz: without it 12 of 13 streaming texts of the reference's
        # own test.wav differed).
        if self._audio_featurizer.use_db_normalization and self.remained_wav.size:
            self.remained_wav = self.remained_wav * db_gain(self.remained_wav, self._audio_featurizer.target_db)
        # frames consumed x hop (10 ms) at the CALLER's sample rate (the reference's constant 160 is 10 ms at 16 kHz,
        # predict.py:275).  DIVERGENCE for streams that are not 16 kHz: the reference resamples its buffered AudioSegment in
        # place (audio_featurizer.py:46-47), so from the second call on it concatenates 16 kHz remainders with raw samples
        # of the caller's rate under the caller's rate label and drops 160 samples per frame of that mixture; here the
        # buffer stays at the caller's rate (featurize() resamples a copy) and the gain above is computed on the
        # un-resampled samples.  16 kHz streams -- the only ones the reference handles consistently -- are identical.
        self.remained_wav = self.remained_wav[int(round(sample_rate * 0.010)) * x_chunk.shape[1]:]

        decoding_chunk_size, context, subsampling = 16, 7, 4
        cached_feature_num = context - subsampling
        decoding_window = (decoding_chunk_size - 1) * subsampling + context
        stride = subsampling * decoding_chunk_size
        num_frames = self.cached_feat.shape[1]
        if num_frames < decoding_window and not is_end:
            return None
        if num_frames < context:
            return None
        left_frames = context if is_end else decoding_window
        score, text, end = None, None, None
        for cur in range(0, num_frames - left_frames + 1, stride):
            end = min(cur + decoding_window, num_frames)
            x = self.cached_feat[:, cur:end, :]
            if self.configs.use_model == "deepspeech2":
                probs_np, lens = self.predictor.predict_chunk_deepspeech(x_chunk=x)
                probs = torch.from_numpy(probs_np)
            else:
                required_cache_size = decoding_chunk_size * -1
                if self.predictor._stream is None:
                    raise NotImplementedError(f"streaming of {self.configs.use_model} is not built yet")
                probs = self.predictor._stream.encode_chunk(x, required_cache_size)  # device tensor [1,c,V]
                lens = np.array([probs.shape[1]])
            if self.configs.decoder == "ctc_beam_search":
                score, text = self.beam_search_decoder.decode_chunk(probs=probs, logits_lens=lens)
            else:
                score, text, self.greedy_last_max_prob_list, self.greedy_last_max_index_list = greedy_decoder_chunk(
                    probs_seq=probs[0], vocabulary=self._text_featurizer.vocab_list,
                    last_max_index_list=self.greedy_last_max_index_list,
                    last_max_prob_list=self.greedy_last_max_prob_list)
        self.cached_feat = self.cached_feat[:, end - cached_feature_num:, :]
        if use_pun or is_itn:
            raise NotImplementedError("punctuation / ITN are outside the hot path")
        return {"text": text, "score": score}

    def reset_stream(self):
        """predict.py:340-347"""
        self.predictor.reset_stream()
        self.remained_wav = None
        self.cached_feat = None
        self.greedy_last_max_prob_list = None
        self.greedy_last_max_index_list = None
        if self.configs.decoder == "ctc_beam_search" and self.beam_search_decoder is not None:
            self.beam_search_decoder.reset_decoder()
