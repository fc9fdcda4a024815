#!/bin/bash
# Per-launch layer table of one UNet step under rocprofv3 (run on the GPU box):  tools/lp.sh TAG LATENT BATCH [--graph]
# -> gpurun_out/r04/layers_TAG.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; lat=$2; b=$3; shift 3
O=$R/gpurun_out/r04/lp_$tag
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/layer_profile.py --latent $lat --batch $b "$@" --dump $O/prog.json > $O.log 2>&1 || { tail -20 $O.log; exit 1; }
python3 $R/tools/layer_profile.py --join $O > $R/gpurun_out/r04/layers_$tag.txt 2>&1
head -3 $R/gpurun_out/r04/layers_$tag.txt; tail -2 $R/gpurun_out/r04/layers_$tag.txt
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
