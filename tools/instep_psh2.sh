#!/bin/bash
# In-step choice among the pre-split F16X2 tiles: every shape on its 2nd / 3rd isolated candidate (tools/psh2_var/v2.json, v3.json),
# per-launch profiles of the step under each table on one box, per-shape winners merged into gpurun_out/plans_ps_h2_instep.json
R=$GRAFT_REPO_ROOT
cd $R
python tools/merge_plans.py --extract ps_f16x2 gpurun_out/plans_ps_h2_instep.json
for lat in 64 32; do
  TABLE_ENV=LDMK_PS_H2_TABLE LAT=$lat bash tools/layer_multi.sh tools/psh2_var/v2.json tools/psh2_var/v3.json
  cd $R
  python3 tools/instep_tune.py pick gpurun_out/plans_ps_h2_instep.json gpurun_out/instep$lat | tee gpurun_out/instep_psh2_pick_$lat.txt
done
one() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for lat in 64 32; do for t in default gpurun_out/plans_ps_h2_instep.json default gpurun_out/plans_ps_h2_instep.json; do
  echo "== latent $lat LDMK_PS_H2_TABLE=$t"
  if [ $t = default ]; then one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30
  else LDMK_PS_H2_TABLE=$t one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30; fi
done; done
