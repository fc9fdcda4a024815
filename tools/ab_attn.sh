#!/bin/bash
# A/B on ONE box of the self-attention kernels alone: f32 matrix cores, bf16x3 (in-loop split; K / V pre-pass), f16x2
for m in f32 x3 x3p h2; do echo "== $m"; python3 tools/attn_bench.py --mode $m; done
