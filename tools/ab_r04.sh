#!/bin/bash
# A/B of the round-4 switches on ONE box: 32x32x3 step and the 128-frame clip (DDIM-20) with / without the pre-split GEMM tiles
# and the pre-split attention
one() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], (d.get('clip') or {}).get('seconds'))"; }
for ps in 1 0; do for ap in 1 0; do
  echo "== LDMK_PS=$ps LDMK_ATTN_PRESPLIT=$ap  latent 32"
  LDMK_PS=$ps LDMK_ATTN_PRESPLIT=$ap one --latent 32 --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30
done; done
for ps in 1 0; do for ap in 1 0; do
  echo "== LDMK_PS=$ps LDMK_ATTN_PRESPLIT=$ap  clip DDIM-20 (policy job)"
  LDMK_PS=$ps LDMK_ATTN_PRESPLIT=$ap one --latent 32 --no-cpu-baseline --no-secondary --no-extras --steps 3 --warmup 1 --clip-steps 20 --clip-policy job
done; done
