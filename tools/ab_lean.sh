#!/bin/bash
# A/B on one box: LDMK_PS_LEAN=0/1 -- the general transposed epilogue of the pre-split tiles everywhere / the lean form (no row or
# column predicates, no loads of absent operands, no select around the folded-LayerNorm arithmetic) on the GEGLU and QKV projections
out=${1:-gpurun_out/r5_ab_lean.txt}
: > $out
python -m pytest tests/test_f16x2_gpu.py tests/test_ps_conv_gpu.py -q -x > gpurun_out/r5_lean_tests.log 2>&1; tail -2 gpurun_out/r5_lean_tests.log >> $out
for v in 0 1; do
  echo "== ps_bench --h2, LDMK_PS_LEAN=$v (GEGLU / QKV rows)" >> $out
  LDMK_PS_LEAN=$v python tools/ps_bench.py --h2 2>&1 | grep -A1 "^GEGLU\|^QKV" | grep "best" >> $out
done
for rep in 1 2; do for v in 0 1; do
  echo "== step, LDMK_PS_LEAN=$v (round $rep)" >> $out
  LDMK_PS_LEAN=$v python bench.py --no-cpu-baseline --no-clip --no-extras --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  64x64x4', d['value'], ' 32x32x3', d.get('secondary',{}).get('value'))" >> $out
done; done
cat $out
