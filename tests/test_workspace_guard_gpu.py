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
