"""Many-session streaming recogniser built on session groups (``ppasr_encode_chunk_group``; per-session stream handles for
the families the library has no group call for).

No reference counterpart: PPASR serves one stream per ``PPASRPredictor`` (``predict_stream``, predict.py:232-337, one
global predictor behind its FastAPI / GUI apps).  ``StreamPool`` keeps the per-session state machine of
``predict_stream`` (audio remainder, cached feature frames, 67-frame windows with stride 64 = 16 output frames,
greedy_decoder_chunk's running lists) for N sessions and advances every session that has a full window buffered with ONE
set of kernel launches per round.  Each session's result equals what its own ``PPASRPredictor.predict_stream`` returns
(ctc_greedy decoder)."""
import numpy as np
import torch

from ppasr_amd.data_utils.featurizer import AudioFeaturizer, db_gain, pcm_bytes_to_float
from ppasr_amd.model_utils.conformer.model import make_stream_group

__all__ = ["StreamPool"]

_WINDOW, _STRIDE, _CONTEXT, _KEEP = 67, 64, 7, 3  # predict.py:277-283


class _Session:
    def __init__(self):
        self.remained_wav = None
        self.cached_feat = None
        self.frame_ids = []    # argmax index of every output frame so far
        self.frame_probs = []  # max prob of the non-blank frames
        self.result = None


class StreamPool:
    def __init__(self, model, vocab_list, n_sessions, preprocess_conf=None, max_seconds=200.0, blank_index=0):
        self.model = model
        self.vocab = list(vocab_list)
        self.blank = blank_index
        # (one set of launches per round for plain Conformer handles; per-session stream handles behind the same interface
        #  for the Squeezeformer and the Efficient-Conformer)
        self.group = make_stream_group(model, n_sessions, max_frames=min(model.max_len, int(max_seconds * 25) + 32))
        self.featurizer = AudioFeaturizer(**(preprocess_conf or {}))
        self.sessions = [_Session() for _ in range(n_sessions)]

    def feed(self, session, audio_data, channels=1, samp_width=2):
        """Append PCM bytes or float samples to a session's buffer (what predict_stream does with each packet)."""
        s = self.sessions[session]
        samples = (pcm_bytes_to_float(audio_data, channels, samp_width) if isinstance(audio_data, (bytes, bytearray))
                   else np.asarray(audio_data, np.float32).reshape(-1))
        s.remained_wav = samples if s.remained_wav is None else np.concatenate([s.remained_wav, samples])
        feat = self.featurizer.featurize(s.remained_wav)
        # predict_stream's buffered samples are normalised IN PLACE by the reference's featurize() (ppasr_amd/predict.py
        # explains): the remainder that stays buffered carries this call's gain
        if self.featurizer.use_db_normalization and s.remained_wav.size:
            s.remained_wav = s.remained_wav * db_gain(s.remained_wav, self.featurizer.target_db)
        if feat.shape[0] > 0:
            feat = feat[np.newaxis]
            s.cached_feat = feat if s.cached_feat is None else np.concatenate([s.cached_feat, feat], axis=1)
            hop = int(round(getattr(self.featurizer, "sample_rate", 16000) * 0.010))  # 10 ms at the featurizer's rate
            s.remained_wav = s.remained_wav[hop * feat.shape[1]:]

    def step(self):
        """Advance, as often as possible, every session that holds a full 67-frame window; -> {session: result dict}
        for the sessions that produced new output."""
        updated = {}
        while True:
            ready = [i for i, s in enumerate(self.sessions)
                     if s.cached_feat is not None and s.cached_feat.shape[1] >= _WINDOW]
            if not ready:
                break
            chunks = np.concatenate([self.sessions[i].cached_feat[:, :_WINDOW] for i in ready], axis=0)
            fa, fp = self.group.encode_chunks(ready, chunks)
            fa, fp = fa.cpu().numpy(), fp.cpu().numpy()
            for k, i in enumerate(ready):
                s = self.sessions[i]
                s.frame_ids.extend(fa[k].tolist())
                s.frame_probs.extend(fp[k][fa[k] != self.blank].tolist())
                # windows advance by the stride; predict_stream keeps `end - 3` frames once no full window is left,
                # which is the same position because the next window starts at cur + 64 = end - 3
                s.cached_feat = s.cached_feat[:, _STRIDE:]
                s.result = self._result(s)
                updated[i] = s.result
        return updated

    def finish(self, session):
        """End of a session's audio (``predict_stream(..., is_end=True)``, predict.py:291-298): run the buffered full
        windows, then the last, shorter window if at least 7 frames (the conv front-end's context) are left.
        -> the session's final result dict (None if it never produced output).  The session keeps its state until
        ``reset``."""
        self.step()
        s = self.sessions[session]
        if s.cached_feat is not None and s.cached_feat.shape[1] >= _CONTEXT:
            fa, fp = self.group.encode_chunks([session], s.cached_feat)
            fa, fp = fa.cpu().numpy(), fp.cpu().numpy()
            s.frame_ids.extend(fa[0].tolist())
            s.frame_probs.extend(fp[0][fa[0] != self.blank].tolist())
            s.cached_feat = s.cached_feat[:, s.cached_feat.shape[1] - _KEEP:]
            s.result = self._result(s)
        return s.result

    def _result(self, s):
        hist = np.asarray(s.frame_ids, np.int64)
        keep = np.ones(len(hist), bool)
        keep[1:] = hist[1:] != hist[:-1]
        ids = hist[keep]
        ids = ids[ids != self.blank]
        score = float(sum(s.frame_probs) / len(s.frame_probs)) * 100.0 if s.frame_probs else 0
        text = "".join(self.vocab[i] for i in ids).replace("<space>", " ")
        return {"text": text, "score": score}

    def reset(self, session):
        self.group.reset(session)
        self.sessions[session] = _Session()
