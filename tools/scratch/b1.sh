cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python tools/sample_clip.py --frames 8 --steps 50 --mode autoreg > gpurun_out/b1_before.json 2>&1
python tools/autotune.py --rows --case 32:1 > gpurun_out/autotune_b1.log 2>&1
cp dsml_thesis_amd/igemm_plans.json gpurun_out/igemm_plans_b1.json
python tools/sample_clip.py --frames 8 --steps 50 --mode autoreg > gpurun_out/b1_after.json 2>&1
tail -1 gpurun_out/b1_before.json; tail -1 gpurun_out/b1_after.json
cd /tmp
mkdir -p $R/gpurun_out/lpb1b
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lpb1b -- python3 $R/tools/layer_profile.py --latent 32 --batch 1 --dump $R/gpurun_out/lpb1b/prog.json > $R/gpurun_out/lpb1b.log 2>&1
python3 $R/tools/layer_profile.py --join $R/gpurun_out/lpb1b > $R/gpurun_out/lpb1b_layers.txt 2>&1
tail -2 $R/gpurun_out/lpb1b_layers.txt
