#!/bin/bash
# A/B on ONE box: attn1.to_out on pre-split tiles (the attention writes its result in the PS layout) or not, F16X2 step
one() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for lat in 64 32; do for v in 1 0 1 0; do
  echo "== latent $lat  LDMK_ATTN_PS=$v"; LDMK_ATTN_PS=$v one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30
done; done
