#!/bin/bash
# Round-4 profile set (run on the GPU box through gpurun): one rocprofv3 kernel-stats file PER CONFIG, PMC traffic and
# MFMA-busy passes (counters in their own runs, no tracing), per-launch layer tables (hipGraph replays), the batch-1
# autoregressive step, the bf16 training step.  Outputs under gpurun_out/r04p/; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04p
# two calls of at most 20 minutes each: PART=1 the bench lines (A/B of the arithmetics on one box), PART=2 the profiler passes
PART=${PART:-12}
mkdir -p $O
if [[ $PART == *1* ]]; then
python3 $R/bench.py > $O/r04_bench_default.json 2> $O/bench_default.err
# the same command with every product on the f32 matrix cores (the round-2 arithmetic), for the A/B on one box
LDMK_SPLIT_BF16=0 python3 $R/bench.py --no-cpu-baseline --no-clip --no-extras > $O/r04_bench_default_f32_mfma.json 2>> $O/bench_default.err
# the bf16x3 arithmetic of this round (LDMK_F16X2=0: six bf16 MFMAs per product, pre-split tiles + pre-split attention) and the round-3
# kernels (no pre-split tiles, attention splitting K / V in its key loop) on this box, for the A/B of this round's work
LDMK_F16X2=0 python3 $R/bench.py --no-cpu-baseline --no-clip --no-extras > $O/r04_bench_bf16x3.json 2>> $O/bench_default.err
LDMK_F16X2=0 LDMK_PS=0 LDMK_ATTN_PRESPLIT=0 python3 $R/bench.py --no-cpu-baseline --no-clip --no-extras > $O/r04_bench_round3_kernels.json 2>> $O/bench_default.err
fi
if [[ $PART != *2* ]]; then ls $O; exit 0; fi
B="python3 $R/bench.py --no-secondary --no-cpu-baseline --no-clip --no-extras"
for lat in 64 32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$lat -- $B --latent $lat > $O/r04_bench${lat}_under_rocprof.json 2> $O/ks$lat.log
  python3 $R/tools/summarize_rocprof.py $O/ks$lat $O/r04_bench${lat}_kernel_stats.txt > /dev/null
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf$lat -- $B --latent $lat --steps 3 --warmup 1 --no-graph > /dev/null 2> $O/pf$lat.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw$lat -- $B --latent $lat --steps 3 --warmup 1 --no-graph > /dev/null 2> $O/pw$lat.log
  python3 $R/tools/pmc_traffic.py $O/pf$lat $O/pw$lat 145 latent${lat}_b16 $O/traffic_r04.json > $O/traffic$lat.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pm$lat -- $B --latent $lat --steps 3 --warmup 1 --no-graph > /dev/null 2> $O/pm$lat.log
  python3 $R/tools/pmc_mfma.py $O/pm$lat > $O/r04_pmc_mfma_busy_$lat.txt 2>&1
  mkdir -p $O/lp$lat
  rocprofv3 --kernel-trace --output-format csv -d $O/lp$lat -- python3 $R/tools/layer_profile.py --latent $lat --graph --dump $O/lp$lat/prog.json > $O/lp$lat.log 2>&1
  python3 $R/tools/layer_profile.py --join $O/lp$lat > $O/r04_layers$lat.txt 2>&1
done
mkdir -p $O/lpb1
rocprofv3 --kernel-trace --output-format csv -d $O/lpb1 -- python3 $R/tools/layer_profile.py --latent 32 --batch 1 --graph --dump $O/lpb1/prog.json > $O/lpb1.log 2>&1
python3 $R/tools/layer_profile.py --join $O/lpb1 > $O/r04_layers32_b1.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/bench.py --train --bf16 --latent 64 --batch 16 --steps 3 --warmup 1 > $O/r04_train_bf16_64.json 2> $O/tr.log
python3 $R/tools/summarize_rocprof.py $O/tr $O/r04_train_step_bf16_kernel_stats.txt > /dev/null
# keep the merge small: drop the raw traces
find $O -name "*.csv" -size +2M -delete
find $O -name "*.db" -delete
ls $O
head -3 $O/r04_layers32_b1.txt; cat $O/traffic_r04.json | head -20
