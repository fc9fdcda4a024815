#!/bin/bash
# A/B on ONE box: the F16X2 step with the pre-split tiles for (a) everything in the ps_f16x2 section of igemm_plans.json, (b) the transformer GEMMs only
# (no Winograd / upsampling entries), (c) none
python3 - <<'PY'
import json
t = json.load(open("dsml_thesis_amd/igemm_plans.json"))["ps_f16x2"]
json.dump({k: v for k, v in t.items() if k.endswith(",1")}, open("gpurun_out/ps_h2_tf_only.json", "w"))
PY
one() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for lat in 64; do for t in default gpurun_out/ps_h2_tf_only.json /nonexistent default gpurun_out/ps_h2_tf_only.json; do
  echo "== latent $lat  LDMK_PS_H2_TABLE=$t"
  if [ $t = default ]; then one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30
  else LDMK_PS_H2_TABLE=$t one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30; fi
done; done
