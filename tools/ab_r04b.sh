#!/bin/bash
# A/B on ONE box of the pre-split attention threshold (tokens per sample from which the K / V pre-pass is used) in the
# 64x64x4 and the 32x32x3 step
one() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for lat in 64 32; do for mt in 2048 256 2048 256; do
  echo "== latent $lat  LDMK_ATTN_PRESPLIT_MIN_TOKENS=$mt"
  LDMK_ATTN_PRESPLIT_MIN_TOKENS=$mt one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30
done; done
