#!/bin/bash
# A/B on ONE box of the self-attention kernels alone: the library before this change (tools/bin/libldmk_r04a.so) against the
# current build, bf16x3 with the in-loop K / V split (x3) and with the pre-pass (x3p; one or two 32-query blocks per wave)
echo "== before: x3";  python3 tools/attn_bench.py --mode x3  --lib tools/bin/libldmk_r04a.so
echo "== before: x3p"; python3 tools/attn_bench.py --mode x3p --lib tools/bin/libldmk_r04a.so
echo "== now: x3";     python3 tools/attn_bench.py --mode x3
echo "== now: x3p, one query block per wave";  LDMK_ATTN_QB=1 python3 tools/attn_bench.py --mode x3p
echo "== now: x3p, two query blocks per wave"; LDMK_ATTN_QB=2 python3 tools/attn_bench.py --mode x3p
