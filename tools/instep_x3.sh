#!/bin/bash
# In-step selection among the bf16x3 plans (GPU box): isolated sweep with runners-up -> variant tables (every shape on its 2nd / 3rd
# isolated candidate) -> per-launch profiles of the step under each -> per-shape pick.  tools/instep_x3.sh LATENT [BATCH] [RANKS]
# Result: gpurun_out/plans_x3_instep.json (+ logs); tools/merge_plans.py --replace bf16x3 puts it into the plan file if it wins end to end.
R=$GRAFT_REPO_ROOT
lat=$1; bat=${2:-16}; ranks=${3:-3}
cd $R
[ -f gpurun_out/plans_x3_instep.json ] || python tools/merge_plans.py --extract bf16x3 gpurun_out/plans_x3_instep.json
cp gpurun_out/plans_x3_instep.json gpurun_out/plans_x3_iso.json
python3 tools/autotune.py --x3 --case $lat:$bat --x3-out gpurun_out/plans_x3_iso.json > gpurun_out/tune_x3_iso_$lat.txt 2>&1
python3 tools/instep_tune.py variants gpurun_out/plans_x3_instep.json gpurun_out/plans_x3_iso.json.cands.json gpurun_out/varx$lat $ranks
python tools/merge_plans.py --replace bf16x3 gpurun_out/plans_x3_instep.json       # t0 = the incumbent
TABLE_ENV=LDMK_X3_TABLE LAT=$lat BATCH=$bat bash tools/layer_multi.sh $(for r in $(seq 2 $ranks); do echo gpurun_out/varx$lat/v$r.json; done)
cd $R
python3 tools/instep_tune.py pick gpurun_out/plans_x3_instep.json gpurun_out/instep$lat | tee gpurun_out/instep_x3_pick_$lat.txt
