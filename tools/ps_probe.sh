#!/bin/bash
# needs a probe build of the library: LDMK_HIPCC_FLAGS=-DLDMK_PS_PROBES python -m dsml_thesis_amd.build (the shipped build ignores these variables)
# what-if probes of the pre-split GEMM (LDMK_PS_DEBUG, csrc/igemm_ps.hip): 0 = real, 1 = DMA issued but dropped (no memory traffic),
# 2 = no DMA instructions, 4 = no matrix instructions, 6 = neither (fragment reads + barriers + epilogue only), 16 = no epilogue
for d in ${PROBES:-0 1 2 4 6 16}; do
  echo "=== LDMK_PS_DEBUG=$d"
  LDMK_PS_DEBUG=$d timeout -k 10 200 python tools/ps_bench.py --probe 2>&1 | grep -v amdgpu.ids
done
