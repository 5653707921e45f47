#!/bin/bash
# In-step same-box A/B of an environment switch: tools/experimental/ab_env.sh VAR "v1 v2 ..." [rounds] [extra bench args]
# Alternates the arms, prints ms_per_step per run.  (bench.py without the CPU baseline / pipe probe / optimizer legs: only the timed steps.)
VAR=$1; VALS=$2; ROUNDS=${3:-2}; shift 3
for r in $(seq 1 $ROUNDS); do
  for v in $VALS; do
    out=$(env $VAR=$v python bench.py --steps 6 --warmup 2 --cpu-baseline off --pipe-probe off --optimizer off --fp32-tower-leg off --other-configs off "$@" 2>/dev/null | tail -1)
    ms=$(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], d['roofline']['frac'])" "$out")
    echo "$VAR=$v round $r: $ms"
  done
done
