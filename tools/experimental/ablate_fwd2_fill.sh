#!/bin/bash
# Timing-only builds: the tile's products and LDS reads with N plain + M transcendental independent fillers per MFMA gap instead of the softmax.
#   bash tools/ablate_fwd2_fill.sh "0 0" "2 0" "4 0" "0 1" "0 2" "2 1" "3 1"
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for pair in "$@"; do
  set -- $pair
  touch $R/tools/experimental/attention_fwd2.hip
  make -C $R/tools/experimental FLAGS_attention_fwd2="-DF2_ABL=${3:-1} -DF2_FILL_PLAIN=$1 -DF2_FILL_EXP=$2" > /tmp/make_f2.log 2>&1 || { tail -5 /tmp/make_f2.log; exit 1; }
  echo "== plain $1 exp $2 abl ${3:-1}"
  timeout -k 10 120 python3 $R/tools/experimental/time_attn_fwd2.py | sed -n 4p
done
