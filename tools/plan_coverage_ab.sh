#!/bin/bash
# A/B on ONE box: what a job of B samples runs at with (a) the nearest tuned plan carried whatever the distance + the F16X2 rule for
# untabled long-K convolutions (round 5 default), (b) without that rule, (c) the 2x rule of rounds 1-4 (no plan beyond a factor 2 of a
# tuned row count: the f32 program), and for B <= 4 at 32x32 the small-batch route against the batched program.
one() { python bench.py --no-extras --no-clip --no-cpu-baseline --no-secondary --steps 20 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(f\"{d['value']:10.1f} sample-steps/s  {d['ms_per_step']:8.3f} ms/step\")"; }
for cfg in "64 2" "64 4" "64 6" "64 16" "64 48" "32 5" "32 7" "32 16" "32 24" "32 48"; do
  set -- $cfg
  echo "== latent $1 B = $2"
  echo -n "  nearest plan + conv rule : "; one --latent $1 --batch $2
  echo -n "  nearest plan, no rule    : "; LDMK_H2_CONV_RULE=0 one --latent $1 --batch $2
  echo -n "  2x rule (rounds 1-4)     : "; LDMK_PLAN_MAX_RATIO=2 LDMK_H2_CONV_RULE=0 one --latent $1 --batch $2
done
for b in 3 4; do
  echo "== latent 32 B = $b"
  echo -n "  small-batch route (default, <= 4096 rows) : "; one --latent 32 --batch $b
  echo -n "  batched F16X2 program (LDMK_SMALL_ROWS=2048): "; LDMK_SMALL_ROWS=2048 one --latent 32 --batch $b
done
