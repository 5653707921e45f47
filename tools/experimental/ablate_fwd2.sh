#!/bin/bash
# Timing-only builds of the second-generation attention forward (GPU box only; results are wrong by construction): which part of a tile costs what.
#   bash tools/ablate_fwd2.sh 0 1 2 3 4 8 12 15
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for abl in "$@"; do
  touch $R/tools/experimental/attention_fwd2.hip
  make -C $R/tools/experimental FLAGS_attention_fwd2="-DF2_ABL=$abl" > /tmp/make_f2_$abl.log 2>&1 || { tail -5 /tmp/make_f2_$abl.log; exit 1; }
  echo "== F2_ABL=$abl"
  timeout -k 10 120 python3 $R/tools/experimental/time_attn_fwd2.py | tail -3
done
