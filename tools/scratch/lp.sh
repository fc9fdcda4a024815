cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lat in 64 32; do
  rm -rf $R/gpurun_out/lp${lat}b; mkdir -p $R/gpurun_out/lp${lat}b
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lp${lat}b -- python3 $R/tools/layer_profile.py --latent $lat --dump $R/gpurun_out/lp${lat}b/prog.json > $R/gpurun_out/lp${lat}b.log 2>&1
  python3 $R/tools/layer_profile.py --join $R/gpurun_out/lp${lat}b > $R/gpurun_out/lp${lat}b_layers.txt 2>&1
  find $R/gpurun_out/lp${lat}b -name "*.csv" -size +20M -delete
done
tail -3 $R/gpurun_out/lp64b_layers.txt
