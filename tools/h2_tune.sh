#!/bin/bash
# F16X2 plan sweep + A/B + layer table (GPU box).  Isolated sweep of the F16X2 tiles against the plans on record -> gpurun_out/plans_h2.json,
# then the step with / without that table on this one box, then the per-launch table of the 64x64x4 step.
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r04
rm -f gpurun_out/plans_h2.json
python3 tools/autotune.py --h2 --fresh --case 64:16 --case 32:16 --x3-out gpurun_out/plans_h2.json > gpurun_out/tune_h2.txt 2>&1 || { tail -20 gpurun_out/tune_h2.txt; exit 1; }
tail -3 gpurun_out/tune_h2.txt
one() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for lat in 64 32; do
  for t in /nonexistent gpurun_out/plans_h2.json /nonexistent gpurun_out/plans_h2.json; do
    echo "== latent $lat  LDMK_H2_TABLE=$t"; LDMK_H2_TABLE=$t one --latent $lat --no-cpu-baseline --no-secondary --no-clip --no-extras --steps 30
  done
done > gpurun_out/ab_h2_table.log 2>&1
cat gpurun_out/ab_h2_table.log
bash tools/lp.sh h2_64 64 16 --graph
