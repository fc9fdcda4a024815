#!/bin/bash
# Round-5 profile set (run on the GPU box through gpurun): one rocprofv3 kernel-stats file PER CONFIG, PMC traffic (by kernel family)
# and MFMA-busy passes (counters in their own runs, no tracing), per-launch layer tables (hipGraph replays), the batch-1
# autoregressive step, the bf16 training step.  Outputs under gpurun_out/r05p/; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
# calls of at most 20 minutes each: PART=1 the bench lines, PART=2 the profiler passes at 64x64x4, PART=3 those at 32x32x3 + batch 1 + training
PART=${PART:-123}
mkdir -p $O
if [[ $PART == *1* ]]; then
python3 $R/bench.py > $O/r05_bench_default.json 2> $O/bench_default.err
fi
B="python3 $R/bench.py --no-secondary --no-cpu-baseline --no-clip --no-extras"
prof() {
  lat=$1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$lat -- $B --latent $lat > $O/r05_bench${lat}_under_rocprof.json 2> $O/ks$lat.log
  python3 $R/tools/summarize_rocprof.py $O/ks$lat $O/r05_bench${lat}_kernel_stats.txt > /dev/null
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf$lat -- $B --latent $lat --steps 3 --warmup 1 --no-graph > /dev/null 2> $O/pf$lat.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw$lat -- $B --latent $lat --steps 3 --warmup 1 --no-graph > /dev/null 2> $O/pw$lat.log
  algo=$([ $lat == 64 ] && echo 2403000000 || echo 1075000000)
  python3 $R/tools/pmc_traffic.py $O/pf$lat $O/pw$lat latent${lat}_b16 $O/traffic_r05.json $algo > $O/traffic$lat.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pm$lat -- $B --latent $lat --steps 3 --warmup 1 --no-graph > /dev/null 2> $O/pm$lat.log
  python3 $R/tools/pmc_mfma.py $O/pm$lat > $O/r05_pmc_mfma_busy_$lat.txt 2>&1
  mkdir -p $O/lp$lat
  rocprofv3 --kernel-trace --output-format csv -d $O/lp$lat -- python3 $R/tools/layer_profile.py --latent $lat --graph --dump $O/lp$lat/prog.json > $O/lp$lat.log 2>&1
  python3 $R/tools/layer_profile.py --join $O/lp$lat > $O/r05_layers$lat.txt 2>&1
}
if [[ $PART == *2* ]]; then prof 64; fi
if [[ $PART == *3* ]]; then
prof 32
mkdir -p $O/lpb1
rocprofv3 --kernel-trace --output-format csv -d $O/lpb1 -- python3 $R/tools/layer_profile.py --latent 32 --batch 1 --graph --dump $O/lpb1/prog.json > $O/lpb1.log 2>&1
python3 $R/tools/layer_profile.py --join $O/lpb1 > $O/r05_layers32_b1.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/bench.py --train --bf16 --latent 64 --batch 16 --steps 3 --warmup 1 > $O/r05_train_bf16_64.json 2> $O/tr.log
python3 $R/tools/summarize_rocprof.py $O/tr $O/r05_train_step_bf16_kernel_stats.txt > /dev/null
fi
# keep the merge small: drop the raw traces
find $O -name "*.csv" -size +2M -delete
find $O -name "*.db" -delete
ls $O
cat $O/traffic_r05.json 2>/dev/null | head -60
