import sys, torch
sys.path.insert(0, '/root/repo')
from ppasr_amd.model_utils.conformer.model import ConformerModel
from ppasr_amd.utils.synth import DEFAULT_VOCAB_SIZE, conformer_state_dict, synth_features
V = DEFAULT_VOCAB_SIZE
conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12, cnn_module_kernel=15)
sd = conformer_state_dict(vocab_size=V, num_blocks=12, seed=1234)
m = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device="cuda:0")
x, _ = synth_features(1, 67, seed=5)
c = torch.from_numpy(x).cuda()
s = m.new_stream()
for _ in range(20):
    s.encode_chunk(c, -16, want_probs=False, want_frames=True)
torch.cuda.synchronize()
