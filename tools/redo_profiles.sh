#!/bin/bash
# re-run single pieces of tools/collect_profiles.sh (after a tool fix): the 32x32x3 layer table
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03q
mkdir -p $O/lp32 && rm -rf $O/lp32/*
rocprofv3 --kernel-trace --output-format csv -d $O/lp32 -- python3 $R/tools/layer_profile.py --latent 32 --graph --dump $O/lp32/prog.json > $O/lp32.log 2>&1
python3 $R/tools/layer_profile.py --join $O/lp32 > $O/r03_layers32.txt 2>&1
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
head -3 $O/r03_layers32.txt; tail -1 $O/r03_layers32.txt
