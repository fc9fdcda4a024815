#!/bin/bash
# A/B on ONE box of the arithmetic switches: F16X2 (default), F16X2 without its pre-split tiles, bf16x3 (round-4 kernels)
one() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['fp32_equivalent_tflops'] if 'fp32_equivalent_tflops' in d['roofline'] else '')"; }
for lat in 64 32; do
  echo "== latent $lat  default (F16X2)";              one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30
  echo "== latent $lat  LDMK_PS_H2_TABLE=/nonexistent (no pre-split tiles in F16X2)"; LDMK_PS_H2_TABLE=/nonexistent one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30
  echo "== latent $lat  LDMK_F16X2=0 (bf16x3)";        LDMK_F16X2=0 one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30
done
