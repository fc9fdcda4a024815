cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in "" "--bf16"; do
  tag=f32; [ -n "$mode" ] && tag=bf16
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tp_$tag -- python3 $R/bench.py --train $mode --steps 5 --warmup 2 --latent 32 > /dev/null 2> $R/gpurun_out/tp_$tag.log
  python3 $R/tools/summarize_rocprof.py $R/gpurun_out/tp_$tag $R/gpurun_out/tp_${tag}_stats.txt | head -24
  find $R/gpurun_out/tp_$tag -name "*.csv" -size +2M -delete
done
