"""Drop-in for ``ppasr/infer_utils/inference_predictor.py`` (``InferencePredictor``): same constructor
arguments, methods and attributes; the paddle.inference handles (named inputs ``speech``,
``speech_lengths``, ``offset``, ``required_cache_size``, ``att_cache``, ``cnn_cache``, :80-97) are
replaced by the HIP library, and the streaming caches stay on the device (``att_cache`` / ``cnn_cache``
/ ``offset`` are materialised in the reference layouts only when read).

numpy in, numpy out, like the reference (:103-145, :184-212).  ``predict_device`` /
``predict_greedy`` are the zero-copy entry points.
"""
import os

import numpy as np
import torch

from ppasr_amd import _lib
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.utils.checkpoint import find_state_dict, load_state_dict

__all__ = ["InferencePredictor"]


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class InferencePredictor:
    def __init__(self, configs, use_model, streaming=True, model_dir="models/conformer_streaming_fbank/infer/",
                 use_gpu=True, use_tensorrt=False, gpu_mem=1000, num_threads=10, state_dict=None, vocab_size=None,
                 device="cuda:0"):
        if not use_gpu:
            raise Exception("ppasr_amd has no CPU path (use_gpu=False is not supported)")
        self.configs = configs
        self.use_model = use_model
        self.streaming = streaming
        if use_model not in ("conformer", "efficient_conformer", "squeezeformer", "deepspeech2"):
            raise Exception(f"没有该模型：{use_model}")  # SUPPORT_MODEL, ppasr/__init__.py:3
        if state_dict is None:
            if not os.path.exists(model_dir):
                raise Exception("模型文件不存在，请检查%s是否存在！" % model_dir)
            state_dict = load_state_dict(find_state_dict(model_dir))
        enc = _get(configs, "encoder_conf", {})
        enc = dict(enc) if isinstance(enc, dict) else dict(vars(enc))
        pre = _get(configs, "preprocess_conf", {})
        input_dim = int(_get(pre, "n_mels", 80))
        if vocab_size is None:
            key = "decoder.ctc_lo.bias" if use_model == "deepspeech2" else "ctc.ctc_lo.bias"
            vocab_size = int(state_dict[key].shape[0])
        # model factory (trainer.py:172-210)
        if use_model == "conformer":
            cls = ConformerModel
        elif use_model == "efficient_conformer":
            from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel as cls
        elif use_model == "squeezeformer":
            from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel as cls
        else:
            from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model as cls
        self.model = cls(input_dim, vocab_size, streaming=streaming, encoder_conf=enc, state_dict=state_dict,
                         device=device)
        # (a configuration whose forward_chunk the library does not build -- input_layer: linear, an Efficient-Conformer
        #  behind conv2d6 / conv2d8 -- still serves predict(); predict_chunk_conformer then raises NotImplementedError)
        self._stream = None
        if streaming and "former" in use_model:
            try:
                self._stream = self.model.new_stream()
            except _lib.PPASRHipError as e:
                # only "this route builds no stream handles"; a failed cache allocation (PPASR_EHIP) must surface here
                if e.status != _lib.PPASR_EUNSUPPORTED:
                    raise
        self.output_state_h = None
        self.output_state_c = None

    # ---- reference attributes, materialised lazily from the device state ----
    @property
    def offset(self):
        return np.array([self._stream.offset if self._stream else 0], dtype=np.int32)

    @property
    def att_cache(self):
        if self._stream is None or self._stream.cache_frames == 0:
            return np.zeros([0, 0, 0, 0], dtype=np.float32)
        return self._stream.export_caches()[0].cpu().numpy()

    @property
    def cnn_cache(self):
        if self._stream is None or self._stream.offset == 0:
            return np.zeros([0, 0, 0, 0], dtype=np.float32)
        return self._stream.export_caches()[1].cpu().numpy()

    # ---- full utterance ----
    def predict_device(self, speech, speech_lengths):
        """-> probs [B,T',V] device tensor (no host round trip)."""
        return self.model.get_encoder_out(speech, speech_lengths)

    def predict_greedy(self, speech, speech_lengths, trim_to_length=False):
        """Fused encoder + greedy decode: -> (tokens, n_tokens, score) device tensors."""
        if self.use_model == "deepspeech2":
            from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decode_ids
            probs, out_lens, _, _ = self.model._run(speech, speech_lengths)
            t, n, s, _, _ = greedy_decode_ids(probs, out_lens.to(torch.int32) if trim_to_length else None)
            return t, n, s
        return self.model.encode_greedy(speech, speech_lengths, trim_to_length=trim_to_length)

    def predict(self, speech, speech_lengths):
        """inference_predictor.py:103-145: probs [B,T',V] numpy.  (The reference's exported streaming
        graph runs the utterance as one chunk with empty caches and no mask, :127-137 — identical to
        get_encoder_out for un-padded input; padded batches use the masked batch path here.)"""
        if self.streaming and self._stream is not None:
            self.reset_stream()
        return self.predict_device(speech, speech_lengths).cpu().numpy()

    # ---- streaming ----
    def predict_chunk_conformer(self, x_chunk, required_cache_size):
        """inference_predictor.py:184-212 -> probs [1,c,V] numpy; caches/offset advance on the device."""
        if not ("former" in self.use_model and self.streaming):
            raise Exception(f"当前模型不支持该方法，当前模型为：{self.use_model}")
        if self._stream is None:
            raise NotImplementedError(f"forward_chunk of {self.use_model} is not built yet (NOTES.md §7)")
        return self._stream.encode_chunk(np.asarray(x_chunk, np.float32), int(required_cache_size)).cpu().numpy()

    def predict_chunk_deepspeech(self, x_chunk):
        """inference_predictor.py:147-182 -> (probs [B,c,V], lens [B]) numpy; the LSTM states
        (output_state_h / output_state_c, [layers,B,rnn_size]) stay on the device between calls."""
        if not (self.use_model == "deepspeech2" and self.streaming):
            raise Exception(f"当前模型不支持该方法，当前模型为：{self.use_model}")
        x = np.asarray(x_chunk, np.float32)
        lens = np.full(x.shape[0], x.shape[1], np.int64)
        probs, out_lens, self.output_state_h, self.output_state_c = self.model.get_encoder_out_chunk(
            x, lens, self.output_state_h, self.output_state_c)
        return probs.cpu().numpy(), out_lens.cpu().numpy()

    def reset_stream(self):
        """inference_predictor.py:215-220"""
        self.output_state_h = None
        self.output_state_c = None
        if self._stream is not None:
            self._stream.reset()
