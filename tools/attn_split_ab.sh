#!/bin/bash
# A/B on one box, the F16X2 attention's key loop at 4096 tokens (QB = 2):
#   LDMK_ATTN_PIPE=0  phase-separated loop (Q K^T, softmax, P V one after the other; round 4 + the integer-grid maximum)
#   LDMK_ATTN_PIPE=1  pipelined: the MFMAs of blocks b - 1 / b + 1 between slices of block b's VALU work -- same bits as 0
#   LDMK_ATTN_PIPE=2  pipelined + lazy running maximum (shipped)
out=${1:-gpurun_out/r5_ab_attn_pipe.txt}
: > $out
for p in 0 1 2; do
  echo "== LDMK_ATTN_PIPE=$p, attention alone, latent 64" >> $out
  LDMK_ATTN_PIPE=$p python tools/attn_bench.py --mode h2 --latent 64 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
for rep in 1 2; do
  for p in 0 1 2; do
    echo "== step, LDMK_ATTN_PIPE=$p (round $rep)" >> $out
    LDMK_ATTN_PIPE=$p python bench.py --no-cpu-baseline --no-clip --no-extras --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  64x64x4', d['value'], ' 32x32x3', d.get('secondary',{}).get('value'))" >> $out || exit 1
  done
done
cat $out
