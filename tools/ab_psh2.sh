#!/bin/bash
# A/B on ONE box: the F16X2 step with / without the pre-split tiles in that arithmetic (igemm_plans_ps_h2.json)
one() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for lat in 64 32; do for t in default /nonexistent default /nonexistent; do
  echo "== latent $lat  LDMK_PS_H2_TABLE=$t"
  if [ $t = default ]; then one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30
  else LDMK_PS_H2_TABLE=$t one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30; fi
done; done
