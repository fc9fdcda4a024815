import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_tf_model
from conftest import rnd
m = make_tf_model(gain=0.25, seq_len=3)
audio = rnd(75, 3, 768).cuda()
idx = torch.tensor([[0,0,1],[0,1,2],[1,2,2]], device="cuda")
a = [m.cond_stage_model_2(audio[idx]).clone() for _ in range(4)]
print("audio att equal:", [torch.equal(a[0], x) for x in a[1:]], [(a[0]-x).abs().max().item() for x in a[1:]])
masked = torch.tanh(rnd(76, 3, 3, 128, 128)).cuda()
e = [m.encode_first_stage(masked).clone() for _ in range(4)]
print("encode equal:", [torch.equal(e[0], x) for x in e[1:]], [(e[0]-x).abs().max().item() for x in e[1:]])
z = rnd(5, 3, 3, 32, 32).cuda()
d = [m.first_stage_model.decode(z, force_not_quantize=True).clone() for _ in range(3)]
print("decode equal:", [torch.equal(d[0], x) for x in d[1:]], [(d[0]-x).abs().max().item() for x in d[1:]])
