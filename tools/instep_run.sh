#!/bin/bash
# One in-step tuning round on the GPU box for latent $1 batch $2 (candidate ranks up to $3, default 3): isolated sweep (candidates) -> variant tables -> per-launch
# profiles of the step under each table -> per-shape pick.  Result: gpurun_out/plans_instep.json (+ log)
R=$GRAFT_REPO_ROOT
lat=$1; bat=${2:-16}; ranks=${3:-3}
cd $R
[ -f gpurun_out/plans_instep.json ] || python tools/merge_plans.py --extract f32 gpurun_out/plans_instep.json
cp gpurun_out/plans_instep.json gpurun_out/plans_iso.json
python3 tools/autotune.py --force --case $lat:$bat --out gpurun_out/plans_iso.json > gpurun_out/tune_iso_$lat.txt 2>&1
python3 tools/instep_tune.py variants gpurun_out/plans_instep.json gpurun_out/plans_iso.json.cands.json gpurun_out/var$lat $ranks
python tools/merge_plans.py --replace f32 gpurun_out/plans_instep.json       # t0 = the incumbent
LAT=$lat BATCH=$bat bash tools/layer_multi.sh gpurun_out/plans_iso.json $(for r in $(seq 2 $ranks); do echo gpurun_out/var$lat/v$r.json; done)
cd $R
python3 tools/instep_tune.py pick gpurun_out/plans_instep.json gpurun_out/instep$lat | tee gpurun_out/instep_pick_$lat.txt
