import torch, math, sys
sys.path.insert(0, '.')
from dsml_thesis_amd import ops
f = ops.timestep_freqs(160)
t = torch.tensor([999], device='cuda')
e = ops.timestep_embedding(t, f, 160).cpu()
fc = f.cpu()
arg = (torch.tensor([999.0]) * fc[21])
print("freq", fc[21].item(), "arg", arg.item(), "dev", e[0,21].item(), "cpu", torch.cos(arg).item())
f2 = torch.exp(-math.log(10000) * torch.arange(0, 80, dtype=torch.float32) / 80)
print((f2-fc).abs().max())
