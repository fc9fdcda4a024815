#!/bin/bash
# Per-launch tables of one UNet step under two plan tables on the same box:  tools/layer_ab.sh TABLE_B.json [latent]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab
lat=${2:-64}
rm -rf $O && mkdir -p $O/a $O/b
rocprofv3 --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/layer_profile.py --latent $lat --dump $O/a/prog.json > $O/a.log 2>&1
python3 $R/tools/layer_profile.py --join $O/a > $O/layers_a.txt 2>&1
export LDMK_PLAN_TABLE=$R/$1
rocprofv3 --kernel-trace --output-format csv -d $O/b -- python3 $R/tools/layer_profile.py --latent $lat --dump $O/b/prog.json > $O/b.log 2>&1
python3 $R/tools/layer_profile.py --join $O/b > $O/layers_b.txt 2>&1
find $O -name "*.csv" -size +2M -delete
find $O -name "*.db" -delete
