#!/bin/bash
# Per-launch tables of one UNet step under the committed plan table (t0) and under each table given (t1, t2, ...), all on
# the same box:  LAT=64 tools/layer_multi.sh A.json B.json ...   -> gpurun_out/instep$LAT/t*/per_key.json, layers.txt
# (TABLE_ENV=LDMK_X3_TABLE: the tables are flat variants of the bf16x3 section instead)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
lat=${LAT:-64}
bat=${BATCH:-16}
O=$R/gpurun_out/instep$lat
rm -rf $O && mkdir -p $O
i=0
for tab in "" "$@"; do
  D=$O/t$i
  mkdir -p $D
  if [ -n "$tab" ]; then export ${TABLE_ENV:-LDMK_PLAN_TABLE}=$R/$tab; fi      # TABLE_ENV=LDMK_X3_TABLE: variants of the bf16x3 table
  rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/tools/layer_profile.py --latent $lat --batch $bat --dump $D/prog.json > $D/run.log 2>&1
  python3 $R/tools/layer_profile.py --join $D > $D/layers.txt 2>&1
  head -1 $D/layers.txt
  find $D -name "*.csv" -delete
  find $D -name "*.db" -delete
  i=$((i+1))
done
