#!/bin/bash
# A/B on ONE box: the 64x64x4 step with one / two 32-query blocks per wave in the pre-split attention
one() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for qb in 2 1 2 1; do
  echo "== latent 64  LDMK_ATTN_QB=$qb"
  LDMK_ATTN_QB=$qb one --latent 64 --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30
done
