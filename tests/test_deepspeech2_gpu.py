"""GPU parity: HIP DeepSpeech2 path (conv + LSTM stack + LayerNorm + CTC) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle.deepspeech2_oracle import DeepSpeech2Oracle
from ppasr_amd.utils.synth import deepspeech2_state_dict, synth_features

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _model(sd, V, L, streaming):
    from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
    return DeepSpeech2Model(80, V, streaming=streaming, encoder_conf=dict(num_rnn_layers=L, rnn_size=1024),
                            state_dict=sd, device="cuda:0")


@pytest.mark.parametrize("streaming", [True, False])
def test_deepspeech2_matches_oracle(streaming):
    V, L = 200, 2
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, streaming=streaming, seed=91, perturb_norm=True)
    x, lens = synth_features(3, 131, lens=[131, 90, 40], seed=92)
    model = _model(sd, V, L, streaming)
    probs, out_lens, fh, fc = model.get_encoder_out_chunk(x, lens)
    torch.cuda.synchronize()
    rp, rl, rh, rc = DeepSpeech2Oracle(sd, L, 1024, streaming).forward(x, lens)
    assert out_lens.cpu().tolist() == rl.tolist()
    assert _rel(probs.cpu().numpy(), rp.numpy()) < TOL
    assert _rel(fh.cpu().numpy(), rh.numpy()) < TOL and _rel(fc.cpu().numpy(), rc.numpy()) < TOL


def test_deepspeech2_streaming_state_carry():
    """predict_chunk_deepspeech semantics (inference_predictor.py:147-182): feeding the final states of one
    call as the initial states of the next matches the oracle doing the same."""
    V, L = 120, 2
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, streaming=True, seed=93)
    x, _ = synth_features(1, 67 * 2, seed=94)
    model = _model(sd, V, L, True)
    oracle = DeepSpeech2Oracle(sd, L, 1024, True)
    h = c = rh = rc = None
    for s in (0, 67):
        chunk = x[:, s:s + 67]
        lens = np.array([67])
        probs, _, h, c = model.get_encoder_out_chunk(chunk, lens, h, c)
        rp, _, rh, rc = oracle.forward(chunk, lens, rh, rc)
        assert _rel(probs.cpu().numpy(), rp.numpy()) < TOL


def test_deepspeech2_config1_shape():
    """BASELINE configs[0]: DeepSpeech2 non-streaming (bidirectional), 5 layers x 1024, B=1, 5 s utterance, greedy."""
    from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decode_ids
    V, L = 4233, 5
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, streaming=False, seed=95)
    x, lens = synth_features(1, 498, seed=96)
    model = _model(sd, V, L, False)
    probs = model.get_encoder_out(x, lens)
    assert tuple(probs.shape) == (1, 123, V)
    rp, _, _, _ = DeepSpeech2Oracle(sd, L, 1024, False).forward(x, lens)
    assert _rel(probs.cpu().numpy(), rp.numpy()) < TOL
    tokens, n, score, _, _ = greedy_decode_ids(probs)
    ref = rp[0].numpy().argmax(1)
    keep = np.r_[True, ref[1:] != ref[:-1]]
    ids = ref[keep]
    assert np.array_equal(tokens[0, :int(n[0])].cpu().numpy(), ids[ids != 0])


def test_deepspeech2_streaming_batch_wavefront_with_state_carry():
    """B >= 4 unidirectional batches take the wavefront path (k_lstm_wave: T + L - 1 launches over (layer, time), the
    LayerNorm + input projection of layers >= 1 folded into the step): ragged lengths, three layers, and the final
    states of one call as the initial states of the next, against the oracle."""
    V, L = 90, 3
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, streaming=True, seed=97, perturb_norm=True)
    model = _model(sd, V, L, True)
    oracle = DeepSpeech2Oracle(sd, L, 1024, True)
    x, _ = synth_features(6, 150, seed=98)
    h = c = rh = rc = None
    for s0, lens in ((0, [75, 75, 60, 33, 9, 75]), (75, [75, 51, 75, 20, 75, 64])):
        chunk = x[:, s0:s0 + 75]
        la = np.array(lens)
        probs, ol, h, c = model.get_encoder_out_chunk(chunk, la, h, c)
        rp, rl, rh, rc = oracle.forward(chunk, la, rh, rc)
        torch.cuda.synchronize()
        assert ol.cpu().tolist() == rl.tolist()
        assert _rel(probs.cpu().numpy(), rp.numpy()) < TOL
        assert _rel(h.cpu().numpy(), rh.numpy()) < TOL and _rel(c.cpu().numpy(), rc.numpy()) < TOL


@pytest.mark.parametrize("streaming,gru,B", [(True, False, 6), (False, False, 1), (True, True, 3), (False, True, 2)])
def test_deepspeech2_rnn_size_2048(streaming, gru, B):
    """rnn_size 2048 (configs/deepspeech2.yml:4 "for big data ... 2048"; the library accepts 1024 and 2048) on every
    recurrence route: the (layer, time) wavefront (B >= 4 unidirectional LSTM), the batched per-step kernels, the
    single-utterance kernels."""
    from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
    V, L, H = 97, 2, 2048
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, rnn_size=H, streaming=streaming, seed=291, perturb_norm=True,
                                use_gru=gru)
    lens = [99] + [int(v) for v in np.linspace(90, 30, B - 1)] if B > 1 else [99]
    x, lens = synth_features(B, 99, lens=lens, seed=292)
    model = DeepSpeech2Model(80, V, streaming=streaming, encoder_conf=dict(num_rnn_layers=L, rnn_size=H, use_gru=gru),
                             state_dict=sd, device="cuda:0")
    probs, out_lens, fh, fc = model.get_encoder_out_chunk(x, lens)
    torch.cuda.synchronize()
    rp, rl, rh, rc = DeepSpeech2Oracle(sd, L, H, streaming, use_gru=gru).forward(x, lens)
    assert out_lens.cpu().tolist() == rl.tolist()
    e = _rel(probs.cpu().numpy(), rp.numpy())
    print(f"H=2048 streaming={streaming} gru={gru} B={B}: probs {e:.2e}")
    assert e < TOL
    assert _rel(fh.cpu().numpy(), rh.numpy()) < TOL


@pytest.mark.parametrize("gru,B", [(False, 40), (False, 70), (True, 6), (True, 45), (True, 100)])
def test_deepspeech2_wavefront_row_tiles_and_gru(gru, B):
    """Round 4: the (layer, time) wavefront takes every batch size -- a workgroup holds 1 / 2 / 4 row tiles of 32 utterances
    (B <= 32 / <= 64 / more) and streams its gate columns' weights once for all of them -- and nn.GRU stacks as well
    (recurrent part kept apart from the input part for the candidate gate).  Ragged lengths, three layers, the final
    states of one call as the initial states of the next, against the oracle (whose cell arithmetic is pinned to the
    reference's CRNNEncoder through the shim)."""
    V, L = 70, 3
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, streaming=True, seed=397, perturb_norm=True, use_gru=gru)
    from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
    model = DeepSpeech2Model(80, V, streaming=True, encoder_conf=dict(num_rnn_layers=L, rnn_size=1024, use_gru=gru),
                             state_dict=sd, device="cuda:0")
    oracle = DeepSpeech2Oracle(sd, L, 1024, True, use_gru=gru)
    x, _ = synth_features(B, 118, seed=398)
    rng = np.random.default_rng(B)
    h = c = rh = rc = None
    for s0 in (0, 59):
        chunk = x[:, s0:s0 + 59]
        la = rng.integers(7, 60, size=B)
        la[0] = 59
        probs, ol, h, c = model.get_encoder_out_chunk(chunk, la, h, c)
        rp, rl, rh, rc = oracle.forward(chunk, la, rh, rc)
        torch.cuda.synchronize()
        assert ol.cpu().tolist() == rl.tolist()
        e = _rel(probs.cpu().numpy(), rp.numpy())
        print(f"gru={gru} B={B}: probs {e:.2e}")
        assert e < TOL
        assert _rel(h.cpu().numpy(), rh.numpy()) < TOL
        if not gru:
            assert _rel(c.cpu().numpy(), rc.numpy()) < TOL


class _persist:
    """PPASR_DS2_PERSIST is read at every call: 0 = the per-step kernels for single utterances."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        import os
        self.old = os.environ.get("PPASR_DS2_PERSIST")
        os.environ["PPASR_DS2_PERSIST"] = "1" if self.on else "0"

    def __exit__(self, *a):
        import os
        if self.old is None:
            os.environ.pop("PPASR_DS2_PERSIST", None)
        else:
            os.environ["PPASR_DS2_PERSIST"] = self.old


@pytest.mark.parametrize("streaming,T,length", [(False, 498, 498), (False, 211, 150), (True, 331, 331), (False, 9, 9)])
def test_single_utterance_persistent_recurrence_equals_per_step_route(streaming, T, length):
    """B = 1, LSTM, 1024 units: a layer's recurrence as ONE persistent launch (W_hh in registers, time steps exchanged through
    tagged 8-byte granules, ds2_kernels.hip k_lstm_persist) against the per-step kernels and the oracle -- probabilities,
    final states, a padded utterance (length < T: the reverse direction starts at length - 1), initial states carried in."""
    V, L = 211, 3
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, streaming=streaming, seed=401, perturb_norm=True)
    model = _model(sd, V, L, streaming)
    x, _ = synth_features(1, T, seed=402)
    lens = np.array([length])
    dirs = 1 if streaming else 2
    rng = np.random.default_rng(5)
    h0 = (0.3 * rng.standard_normal((L * dirs, 1, 1024))).astype(np.float32)
    c0 = (0.3 * rng.standard_normal((L * dirs, 1, 1024))).astype(np.float32)
    outs = []
    for on in (True, False):
        with _persist(on):
            probs, ol, fh, fc = model.get_encoder_out_chunk(x, lens, h0, c0)
            torch.cuda.synchronize()
        outs.append((probs.cpu().numpy(), ol.cpu().numpy(), fh.cpu().numpy(), fc.cpu().numpy()))
    rp, rl, rh, rc = DeepSpeech2Oracle(sd, L, 1024, streaming).forward(x, lens, torch.from_numpy(h0), torch.from_numpy(c0))
    for got in outs:
        assert got[1].tolist() == rl.tolist()
        assert _rel(got[0], rp.numpy()) < TOL and _rel(got[2], rh.numpy()) < TOL and _rel(got[3], rc.numpy()) < TOL
    e = _rel(outs[0][0], outs[1][0])
    print(f"persistent vs per-step: probs {e:.2e} h {_rel(outs[0][2], outs[1][2]):.2e} c {_rel(outs[0][3], outs[1][3]):.2e}")
    assert e < 2e-5 and _rel(outs[0][2], outs[1][2]) < 2e-5 and _rel(outs[0][3], outs[1][3]) < 2e-5
    assert model._h is not None


@pytest.mark.parametrize("streaming,T,length", [(False, 331, 331), (False, 211, 150), (True, 331, 331), (False, 9, 9)])
def test_single_utterance_gru_persistent_recurrence_equals_per_step_route(streaming, T, length):
    """nn.GRU stacks (deepspeech2/encoder.py:36-42, `use_gru`) on the persistent route (k_lstm_persist<GRU>): one launch per
    layer with 3 x 8 weight rows per workgroup in registers, against the per-step kernels (k_gru_step) and the oracle --
    probabilities and final h; the c box is handed through unchanged (encoder.py:95-97)."""
    from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
    V, L = 211, 3
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, streaming=streaming, seed=411, perturb_norm=True, use_gru=True)
    model = DeepSpeech2Model(80, V, streaming=streaming, encoder_conf=dict(num_rnn_layers=L, rnn_size=1024, use_gru=True),
                             state_dict=sd, device="cuda:0")
    x, _ = synth_features(1, T, seed=412)
    lens = np.array([length])
    dirs = 1 if streaming else 2
    rng = np.random.default_rng(6)
    h0 = (0.3 * rng.standard_normal((L * dirs, 1, 1024))).astype(np.float32)
    c0 = (0.3 * rng.standard_normal((L * dirs, 1, 1024))).astype(np.float32)
    outs = []
    for on in (True, False):
        with _persist(on):
            probs, ol, fh, fc = model.get_encoder_out_chunk(x, lens, h0, c0)
            torch.cuda.synchronize()
        outs.append((probs.cpu().numpy(), ol.cpu().numpy(), fh.cpu().numpy(), fc.cpu().numpy()))
    rp, rl, rh, _ = DeepSpeech2Oracle(sd, L, 1024, streaming, use_gru=True).forward(x, lens, torch.from_numpy(h0), torch.from_numpy(c0))
    for got in outs:
        assert got[1].tolist() == rl.tolist()
        assert _rel(got[0], rp.numpy()) < TOL and _rel(got[2], rh.numpy()) < TOL
        assert np.array_equal(got[3], c0)  # handed through
    e = _rel(outs[0][0], outs[1][0])
    print(f"GRU persistent vs per-step: probs {e:.2e} h {_rel(outs[0][2], outs[1][2]):.2e}")
    assert e < 2e-5 and _rel(outs[0][2], outs[1][2]) < 2e-5


@pytest.mark.parametrize("gru", [False, True])
def test_persistent_recurrence_gives_up_when_the_chip_is_partly_held(gru):
    """The persistent launch needs every workgroup of its grid resident at once (256 workgroups; two fit a CU).  With 224
    of the 256 CUs held by another stream's kernel for ~1.5 s (ppasr_debug_occupy_cus) it cannot be: its workgroups spin to
    their bound, raise the give-up flag, and
    the call re-runs on the per-step kernels before it returns -- the result must be the per-step route's, the handle must
    stay off the persistent route for its hold period and come back to it afterwards."""
    import time
    from ppasr_amd import _lib
    from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
    lib = _lib.load()
    V, L, T = 157, 2, 131
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, streaming=False, seed=421, perturb_norm=True, use_gru=gru)
    model = DeepSpeech2Model(80, V, streaming=False, encoder_conf=dict(num_rnn_layers=L, rnn_size=1024, use_gru=gru),
                             state_dict=sd, device="cuda:0")
    x, _ = synth_features(1, T, seed=422)
    lens = np.array([T])
    with _persist(False):
        want = model.get_encoder_out_chunk(x, lens)[0].cpu().numpy()
    with _persist(True):
        free = model.get_encoder_out_chunk(x, lens)[0].cpu().numpy()  # (the chip is free: the persistent route)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        _lib.check(lib.ppasr_debug_occupy_cus(224, 1500, side.cuda_stream))
        time.sleep(0.05)  # (the occupier's workgroups are on their CUs)
        t0 = time.perf_counter()
        held = model.get_encoder_out_chunk(x, lens)[0].cpu().numpy()
        dt = time.perf_counter() - t0
        side.synchronize()
        after = model.get_encoder_out_chunk(x, lens)[0].cpu().numpy()  # inside the hold period: per-step kernels
    print(f"give-up path ({'GRU' if gru else 'LSTM'}): call took {dt * 1e3:.0f} ms with 224 CUs held")
    assert _rel(free, want) < 2e-5
    assert np.array_equal(held, want), "the give-up path must return the per-step route's result"
    assert np.array_equal(after, want)
