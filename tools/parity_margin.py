"""How far inside its tolerance the UNet forward sits: max |eps - reference| on the golden fixtures (g4 FR, g5 TF, g11
north-star shape), with LayerNorm folded through the product (default) and applied in the A staging (LDMK_LN_UNFOLDED=1).

    python tools/parity_margin.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_unet_gpu as T
import inspect
src = inspect.getsource(T)
# re-run the three golden comparisons with a reporting `close`
worst = {}
def close(a, b, rtol, atol):
    a = a.float().cpu(); b = torch.as_tensor(np.asarray(b)).float()
    d = (a - b).abs().max().item(); s = b.abs().max().item()
    worst[len(worst)] = (d, s, atol + rtol * s)
T.close = close
for name in ("test_unet_fr_golden", "test_unet_tf_concat_golden", "test_unet_northstar_64_golden", "test_unet_multi_token_context_vs_oracle"):
    fn = getattr(T, name, None)
    if fn is None:
        continue
    n0 = len(worst)
    fn()
    for k in range(n0, len(worst)):
        d, s, tol = worst[k]
        print(f"{name:28s} max |diff| {d:.3e}  (max |ref| {s:.3f}; allowed ~{tol:.1e})")
'''.replace("ROOT", repr(ROOT))
for env in ({}, {"LDMK_LN_UNFOLDED": "1"}):
    print("LayerNorm", "in the A staging" if env else "folded through the product")
    subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **env), check=False)
