#!/bin/bash
# needs a probe build of the library: LDMK_HIPCC_FLAGS=-DLDMK_PS_PROBES python -m dsml_thesis_amd.build (the shipped build ignores these variables)
# the phase stagger of csrc/igemm_ps.hip (LDMK_PS_STAGGER = 64-cycle units of start delay per 16-deep stage for the workgroups in odd wave slots)
for st in 0 8 16 24 32 48; do
  echo "=== LDMK_PS_STAGGER=$st"
  LDMK_PS_STAGGER=$st timeout -k 10 200 python tools/ps_bench.py --quick 2>&1 | grep -v amdgpu.ids | grep best
done
