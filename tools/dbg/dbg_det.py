import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_tf_model, make_fr_model
from conftest import rnd
from dsml_thesis_amd.engine import GraphedProgram
m = make_tf_model(gain=0.25, seq_len=3)
unet = m.model.diffusion_model
x, t = rnd(1, 1, 3, 32, 32).cuda(), torch.tensor([500]).cuda()
c12, c34 = rnd(2, 1, 1, 1024).cuda(), rnd(3, 1, 6, 32, 32).cuda()
a = unet(x, t, context=c12, c_concat=c34)
b = unet(x, t, context=c12, c_concat=c34)
print("eager vs eager equal:", torch.equal(a, b), (a - b).abs().max().item())
pg = unet.program(1, 32, 32, 1, 6)
g = GraphedProgram(pg.run)
g.replay(); torch.cuda.synchronize()
c = pg.outputs["eps"].clone()
print("eager vs graph equal:", torch.equal(a, c), (a - c).abs().max().item())
g.replay(); torch.cuda.synchronize()
d = pg.outputs["eps"].clone()
print("graph vs graph equal:", torch.equal(c, d), (c - d).abs().max().item())
# per-call divergence search: run eager, snapshot every buffer after each call
import ctypes
def snapshot_run():
    outs = []
    st = torch.cuda.current_stream().cuda_stream
    for fn, args, keep, name in pg.calls:
        fn(*args, st)
        torch.cuda.synchronize()
        if name == "ldmk_igemm":
            n = keep.M * (keep.N // 2 if keep.epi else keep.N)
            buf = (ctypes.c_float * 1)
            t_ = torch.empty(n, device="cuda")
            # copy out through a raw pointer view
            src = torch.cuda.FloatTensor().set_(torch.cuda.FloatStorage._new_with_weak_ptr) if False else None
        outs.append(name)
    return outs
