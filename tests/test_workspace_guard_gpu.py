"""Nothing writes past the workspace size the library asks for: every encode runs with its workspace followed by a guard
region full of a sentinel (and the output buffers followed by one too), for the fused and the split routes, the ragged
mode, streaming chunks and session groups."""
import numpy as np
import pytest
import torch

import test_ragged_gpu as rg
from ppasr_amd.utils.synth import synth_features

pytestmark = pytest.mark.gpu
GUARD = 1 << 20
SENTINEL = 0xA5


def _guarded(nbytes, dev):
    buf = torch.full((nbytes + GUARD,), SENTINEL, dtype=torch.uint8, device=dev)
    return buf


def _intact(buf, nbytes):
    return bool((buf[nbytes:] == SENTINEL).all())


@pytest.mark.parametrize("family", list(rg.FAMILIES))
@pytest.mark.parametrize("B,T,lens", [(1, 67, [67]), (3, 400, [400, 133, 36]), (9, 1000, None)])
def test_encode_stays_inside_its_workspace(family, B, T, lens):
    model, _ = rg.FAMILIES[family](211)
    x, la = synth_features(B, T, lens=lens, seed=B + T)
    need = int(model.lib.ppasr_workspace_bytes(model._h, B, T))
    for mode, skip in ((-1, False), (0, False), (8, False), (-1, True)):
        model.set_ffn_split(mode)
        model.set_skip_padding(skip)
        ws = _guarded(need, model.device)
        key = torch.cuda.current_stream(model.device).cuda_stream
        model._ws = {key: ws[:need]}  # a view of exactly the requested size: the wrapper keeps using it
        probs, logits = model.get_encoder_out(x, la, return_logits=True)
        torch.cuda.synchronize()
        assert _intact(ws, need), (family, mode, skip)
        assert bool(torch.isfinite(probs).all())
    model.set_ffn_split(-1)
    model.set_skip_padding(False)
    model._ws = None


def test_stream_chunk_and_group_stay_inside_their_workspaces():
    from ppasr_amd.model_utils.conformer.model import ConformerStreamGroup
    model, _ = rg.FAMILIES["conformer"](211)
    x, _ = synth_features(1, 67 + 64 * 3, seed=3)
    s = model.new_stream()
    need = int(model.lib.ppasr_chunk_workspace_bytes(model._h, 67))
    ws = _guarded(need, model.device)
    s._ws = ws[:need]
    for k in range(4):
        s.encode_chunk(x[:, 64 * k:64 * k + 67], -16)
    torch.cuda.synchronize()
    assert _intact(ws, need)
    g = ConformerStreamGroup(model, 5, max_frames=256)
    need = int(model.lib.ppasr_group_chunk_workspace_bytes(model._h, 5, 67))
    ws = _guarded(need, model.device)
    key = torch.cuda.current_stream(model.device).cuda_stream
    g._ws = {key: ws[:need]}
    chunks = np.repeat(x[:, :67], 5, axis=0)
    for _ in range(3):
        g.encode_chunks([0, 1, 2, 3, 4], chunks)
    torch.cuda.synchronize()
    assert _intact(ws, need)


@pytest.mark.parametrize("streaming,B", [(True, 1), (True, 6), (False, 3)])
def test_deepspeech2_stays_inside_its_workspace(streaming, B):
    from ppasr_amd.model_utils.deepspeech2.model import DeepSpeech2Model
    from ppasr_amd.utils.synth import deepspeech2_state_dict
    V, L = 60, 3
    sd = deepspeech2_state_dict(vocab_size=V, num_rnn_layers=L, streaming=streaming, seed=5)
    m = DeepSpeech2Model(80, V, streaming=streaming, encoder_conf=dict(num_rnn_layers=L, rnn_size=1024), state_dict=sd,
                         device="cuda:0")
    x, la = synth_features(B, 131, lens=[131] + [int(v) for v in np.linspace(20, 120, B - 1)], seed=B)
    need = int(m.lib.ppasr_ds2_workspace_bytes(m._h, B, 131))
    ws = _guarded(need, m.device)
    m._ws = ws[:need]
    probs, _, h, c = m.get_encoder_out_chunk(x, la)
    torch.cuda.synchronize()
    assert _intact(ws, need) and bool(torch.isfinite(probs).all()) and bool(torch.isfinite(h).all())


@pytest.mark.parametrize("beam,chunks", [(10, 1), (300, 1), (25, 3)])
def test_beam_search_stays_inside_its_state_buffer(beam, chunks):
    from ppasr_amd.decoders import beam_search_decoder as bsd
    rng = np.random.Generator(np.random.PCG64(beam))
    B, T, V = 3, 60, 500
    p = rng.random((B, T * chunks, V)).astype(np.float32) ** 8
    p /= p.sum(-1, keepdims=True)
    dev = torch.device("cuda:0")
    st = bsd._BeamState(B, T * chunks, beam, dev)
    guard = _guarded(st.bytes, dev)
    st.buf = guard[:st.bytes]
    for k in range(chunks):
        tokens, lens, scores, st = bsd.beam_search_ids(torch.from_numpy(p[:, k * T:(k + 1) * T]).cuda(), beam, 0.99, 40, 0,
                                                       state=st)
    torch.cuda.synchronize()
    assert _intact(guard, st.bytes)
    assert int(lens.min()) >= 0
