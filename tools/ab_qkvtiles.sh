#!/bin/bash
# A/B on ONE box: the QKV projection writes the attention's K / V tiles from its epilogue (LDMK_QKV_TILES=1, default) or the attention
# runs its own pre-pass over the fp32 K / V (0)
one() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for lat in 64 32; do for v in 1 0 1 0; do
  echo "== latent $lat  LDMK_QKV_TILES=$v"; LDMK_QKV_TILES=$v one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30
done; done
