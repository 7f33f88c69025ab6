"""GPU: ragged launches through the list of active row blocks (rowblock.h PadSkip::tab, k_block_table) against the padded
grid with early exits (PPASR_BLOCK_TABLE=0; read once at library load: separate processes) -- the same blocks computed by
the same code, only dealt to the chip in another order: bit-identical outputs, for the three *former families, both block
sizes the route rule picks, and a batch whose last block is partial."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from ppasr_amd.utils.synth import conformer_state_dict, efficient_conformer_state_dict, squeezeformer_state_dict, synth_features
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.model_utils.efficient_conformer.model import EfficientConformerModel
from ppasr_amd.model_utils.squeezeformer.model import SqueezeformerModel
V = 89
models = {}
sd = conformer_state_dict(vocab_size=V, num_blocks=3, seed=101, perturb_norm=True)
models["conformer"] = ConformerModel(80, V, streaming=True, encoder_conf=dict(output_size=256, attention_heads=4, linear_units=2048,
                                     num_blocks=3, cnn_module_kernel=15), state_dict=sd, device="cuda:0")
sd = efficient_conformer_state_dict(vocab_size=V, num_blocks=4, seed=102, perturb_norm=True, stride_layer_idx=1, group_layer_idx=(0, 1))
models["efficient"] = EfficientConformerModel(80, V, streaming=True, encoder_conf=dict(output_size=256, attention_heads=4,
    linear_units=2048, num_blocks=4, cnn_module_kernel=15, cnn_module_norm="layer_norm",
    efficient_conf=dict(stride_layer_idx=[1], stride=[2], group_layer_idx=[0, 1], group_size=3, stride_kernel=True)),
    state_dict=sd, device="cuda:0")
sd = squeezeformer_state_dict(vocab_size=V, num_blocks=5, seed=103, perturb_norm=True)
models["squeezeformer"] = SqueezeformerModel(80, V, streaming=True, encoder_conf=dict(encoder_dim=256, output_size=256,
    attention_heads=4, num_blocks=5, reduce_idx=2, recover_idx=4, feed_forward_expansion_factor=8, cnn_module_kernel=31),
    state_dict=sd, device="cuda:0")
out = {}
for B, T in ((16, 607), (40, 607)):   # 16 x 151 frames: 16-row blocks by the rule; 40 x 151: 32-row blocks, last one partial
    rng = np.random.default_rng(B)
    lens = [T] + [int(v) for v in rng.integers(40, T + 1, size=B - 1)]
    x, la = synth_features(B, T, lens=lens, seed=104 + B)
    for name, m in models.items():
        m.set_skip_padding(True)
        m.set_lengths_hint(lens)
        out["%%s_%%d" %% (name, B)] = m.get_encoder_out(x, la, return_logits=True)[1].cpu().numpy()
        m.set_lengths_hint(None)
        m.set_skip_padding(False)
torch.cuda.synchronize()
np.savez(sys.argv[1], **out)
""" % ROOT


def test_block_lists_compute_what_the_padded_grid_computes():
    outs = []
    for flag in ("1", "0"):
        path = f"/tmp/_blocktab_{flag}.npz"
        subprocess.check_call([sys.executable, "-c", CODE, path], env=dict(os.environ, PPASR_BLOCK_TABLE=flag), cwd=ROOT)
        outs.append(np.load(path))
    assert sorted(outs[0].files) == sorted(outs[1].files) and len(outs[0].files) == 6
    for k in outs[0].files:
        a, b = outs[0][k], outs[1][k]
        assert np.isfinite(a).all() and np.abs(a).max() > 0
        assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))
