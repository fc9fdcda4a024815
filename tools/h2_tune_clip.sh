#!/bin/bash
# F16X2 plan sweep for the 128-frame clip batch (32x32, B = 128) merged into a copy of the table, then the clip leg with both tables
R=$GRAFT_REPO_ROOT
cd $R
python tools/merge_plans.py --extract f16x2 gpurun_out/plans_h2_clip.json
python3 tools/autotune.py --h2 --case 32:128 --x3-out gpurun_out/plans_h2_clip.json > gpurun_out/tune_h2_clip.txt 2>&1 || { tail -20 gpurun_out/tune_h2_clip.txt; exit 1; }
tail -2 gpurun_out/tune_h2_clip.txt
one() { python3 bench.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print((d.get('clip') or {}).get('seconds'), (d.get('clip') or {}).get('checksum'))"; }
for t in default gpurun_out/plans_h2_clip.json default gpurun_out/plans_h2_clip.json; do
  echo "== clip DDIM-20  LDMK_H2_TABLE=$t"
  if [ $t = default ]; then one --latent 32 --no-cpu-baseline --no-secondary --no-extras --steps 3 --warmup 1 --clip-steps 20
  else LDMK_H2_TABLE=$t one --latent 32 --no-cpu-baseline --no-secondary --no-extras --steps 3 --warmup 1 --clip-steps 20; fi
done
