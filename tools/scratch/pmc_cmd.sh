cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in 7 8; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_rg_a$cfg -o a -- python3 $R/tools/rgemm_one.py 65536 160 160 $cfg 10 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum -d $R/gpurun_out/pmc_rg_b$cfg -o b -- python3 $R/tools/rgemm_one.py 65536 160 160 $cfg 10 > /dev/null 2>&1
done
ls -R $R/gpurun_out/pmc_rg_a7 | head
