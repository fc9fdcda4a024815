import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_tf_model
from conftest import rnd
from dsml_thesis_amd.ddim import DDIMSampler, C12, C34
m = make_tf_model(gain=0.25, seq_len=3)
s = DDIMSampler(m)
c12, c34 = rnd(2, 1, 1, 1024).cuda(), rnd(3, 1, 6, 32, 32).cuda()
xT = rnd(4, 1, 3, 32, 32).cuda()
cond = {C12: c12, C34: c34}
res = {}
for mode in (False, True, False, True):
    steps = []
    out, inter = s.sample(4, 1, [3, 32, 32], cond, eta=0.0, x_T=xT, verbose=False, use_graph=mode, log_every_t=1)
    res.setdefault(mode, []).append([t.clone() for t in inter["x_inter"]])
for i in range(len(res[False][0])):
    e0, e1, g0, g1 = res[False][0][i], res[False][1][i], res[True][0][i], res[True][1][i]
    print(i, "e0==e1", torch.equal(e0, e1), "g0==g1", torch.equal(g0, g1), "e0==g0", torch.equal(e0, g0),
          (e0 - g0).abs().max().item())
