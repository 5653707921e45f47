#!/bin/bash
# In-step same-box A/B of library builds: tools/experimental/ab_lib.sh "lib1 lib2 ..." [rounds]
B="python bench.py --steps 6 --warmup 2 --cpu-baseline off --pipe-probe off --optimizer off --fp32-tower-leg off --other-configs off"
for r in $(seq 1 ${2:-2}); do for l in $1; do
  MI355_LIB_PATH=$l $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['ms_per_step'])"
done; done
