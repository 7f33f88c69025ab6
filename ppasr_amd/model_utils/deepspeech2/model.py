"""Host-side mirror of ``ppasr/model_utils/deepspeech2/model.py`` (``DeepSpeech2Model``): ``get_encoder_out``
(:62-65) and ``get_encoder_out_chunk`` (:67-72), backed by ``ppasr_ds2_encode`` (include/ppasr_hip.h)."""
import ctypes

import numpy as np
import torch

from ppasr_amd import _lib

__all__ = ["DeepSpeech2Model"]


class DeepSpeech2Model:
    def __init__(self, input_dim, vocab_size, mean_istd_path=None, streaming=True, encoder_conf=None,
                 decoder_conf=None, state_dict=None, device="cuda:0", **_ignored):
        if state_dict is None:
            raise ValueError("state_dict (Paddle-layout parameter dict) is required")
        if not torch.cuda.is_available():
            raise _lib.PPASRHipError("no HIP device visible: ppasr_amd has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.input_dim, self.vocab_size, self.streaming = input_dim, vocab_size, streaming
        conf = dict(encoder_conf or {})
        self.num_rnn_layers = int(conf.get("num_rnn_layers", 5))
        self.rnn_size = int(conf.get("rnn_size", 1024))
        self.use_gru = bool(conf.get("use_gru", False))  # nn.GRU layers (deepspeech2/encoder.py:36-42)
        self.dirs = 1 if streaming else 2
        sd = dict(state_dict)
        keep = []
        blobs = (_lib.WeightBlob * len(sd))()
        for i, (name, arr) in enumerate(sd.items()):
            a = np.ascontiguousarray(arr, dtype=np.float32)
            keep.append(a)
            blobs[i].name = name.encode()
            blobs[i].data_host = a.ctypes.data
            blobs[i].ndim = min(a.ndim, 4)
            for j in range(min(a.ndim, 4)):
                blobs[i].shape[j] = a.shape[j]
        desc = _lib.ModelDesc(_lib.PPASR_MODEL_DEEPSPEECH2, input_dim, vocab_size, self.rnn_size, 0, 0,
                              self.num_rnn_layers, 0, 1 if streaming else 0, 0, -1, -1, -1, 0, 0,
                              1 if self.use_gru else 0)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ppasr_create(ctypes.byref(desc), blobs, len(sd), ctypes.byref(handle)))
        self._h = handle
        self._ws = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self.lib.ppasr_destroy(h)
            self._h = None

    def out_frames(self, T):
        return ((T - 1) // 2 - 1) // 2

    def _run(self, speech, speech_lengths, init_h=None, init_c=None, want_states=False):
        x = torch.as_tensor(speech, dtype=torch.float32).to(self.device).contiguous()
        lens = torch.as_tensor(speech_lengths, dtype=torch.int64).to(self.device).contiguous()
        B, T, _ = x.shape
        Tp = self.out_frames(T)
        need = int(self.lib.ppasr_ds2_workspace_bytes(self._h, B, T))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        probs = torch.empty(B, Tp, self.vocab_size, dtype=torch.float32, device=self.device)
        out_lens = torch.empty(B, dtype=torch.int64, device=self.device)
        S = self.num_rnn_layers * self.dirs
        fh = fc = ih = ic = None
        if want_states:
            fh = torch.empty(S, B, self.rnn_size, dtype=torch.float32, device=self.device)
            fc = torch.empty_like(fh)
        if init_h is not None:
            ih = torch.as_tensor(init_h, dtype=torch.float32).to(self.device).contiguous()
            ic = torch.as_tensor(init_c, dtype=torch.float32).to(self.device).contiguous()
            assert tuple(ih.shape) == (S, B, self.rnn_size)
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self.lib.ppasr_ds2_encode(self._h, x.data_ptr(), lens.data_ptr(), B, T, ptr(ih), ptr(ic),
                                                 probs.data_ptr(), out_lens.data_ptr(), ptr(fh), ptr(fc),
                                                 self._ws.data_ptr(), self._ws.numel(), stream))
        return probs, out_lens, fh, fc

    def get_encoder_out(self, speech, speech_lengths):
        """-> ctc_probs [B, T', V] (device tensor).  deepspeech2/model.py:62-65"""
        return self._run(speech, speech_lengths)[0]

    def get_encoder_out_chunk(self, speech, speech_lengths, init_state_h_box=None, init_state_c_box=None):
        """-> (ctc_probs, eouts_len, final_h_box, final_c_box).  deepspeech2/model.py:67-72"""
        return self._run(speech, speech_lengths, init_state_h_box, init_state_c_box, want_states=True)
