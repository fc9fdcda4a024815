import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import rnd
from dsml_thesis_amd import ops, lib as L
import torch.nn.functional as F
n, cin, cout, h, w = 2, 160, 320, 8, 8
x, wt, b = rnd(10, n, cin, h, w), rnd(11, cout, cin, 3, 3) / np.sqrt(9 * cin), 0.5 + 0.1 * rnd(12, cout)
xd = x.permute(0,2,3,1).contiguous().cuda(); wd = ops.pack_conv3x3(wt.cuda()); bd = b.cuda()
out = torch.empty(n, h, w, cout, device="cuda")
part = torch.zeros(n * h * w // 32, cout, 3, device="cuda")
a = ops.make_igemm_args(n * h * w, cout, 9 * cin, xd, cin, wd, out, cout, h * w, conv=(h, w, h, w, 1, 1, 0), bias=bd)
a.stats_out = part.data_ptr()
ops.igemm(a)
ref = F.conv2d(x, wt, b, padding=1).permute(0,2,3,1).reshape(n*h*w, cout)
print("out err", (out.view(-1, cout).cpu() - ref).abs().max().item())
p2 = torch.zeros_like(part)
L.call("ldmk_gn_partial", out.data_ptr(), cout, n, h*w, p2.data_ptr(), ops.stream())
print("shift err", (part[...,0]-p2[...,0]).abs().max().item(), "sum err", (part[...,1]-p2[...,1]).abs().max().item(), "sq err", (part[...,2]-p2[...,2]).abs().max().item())
print(part[0, :4], p2[0, :4])
cfg, sk = L.C.c_int(0), L.C.c_int(0)
L.load().ldmk_igemm_plan(L.C.byref(a), L.C.byref(cfg), L.C.byref(sk)); print("cfg", cfg.value, sk.value)
