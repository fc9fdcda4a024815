import sys, os, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from dsml_thesis_amd import lib as L, ops
from conftest import rnd
n, tokens, heads = 1, 256, 20
C_ = heads * 32
qkv = (rnd(570, n * tokens, 3 * C_) * 1.2).cuda()
ref = ops.attn_self(qkv, n, tokens, heads, presplit=True)
kv = torch.empty(L.load().ldmk_attn_kv_split_bytes(n, tokens, heads), device="cuda", dtype=torch.uint8)
out = torch.zeros(n * tokens, C_, device="cuda")
ps = ops.ps_empty(n * tokens, C_)
L.call("ldmk_attn_self_x3p_ps", qkv.data_ptr(), kv.data_ptr(), out.data_ptr(), ps.data_ptr(), n, tokens, heads, 32 ** -0.5, ops.stream())
print("out == ref", torch.equal(out, ref))
h, m, l = ops.unpack_ps(ps.cpu(), n * tokens, C_)
rh, rm, rl = ops.unpack_ps(ops.pack_ps(ref).cpu(), n * tokens, C_)
for name, a, b in (("hi", h, rh), ("mid", m, rm), ("lo", l, rl)):
    d = (a != b)
    print(name, int(d.sum()), "mismatches", (a - b).abs().max().item())
s = (h.double() + m.double() + l.double()).float()
print("sum planes == out:", torch.equal(s, out.cpu()), (s - out.cpu()).abs().max().item())
idx = (l != rl).nonzero()[:5]
for i, j in idx.tolist():
    print(i, j, out[i, j].item(), h[i, j].item(), m[i, j].item(), l[i, j].item(), "|", rh[i, j].item(), rm[i, j].item(), rl[i, j].item())
