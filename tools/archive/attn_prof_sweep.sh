#!/bin/bash
# In-kernel stamps of the dK/dV pass under several profiling builds (GPU box only; the library it leaves behind is a profiling build).
#   bash tools/attn_prof_sweep.sh 16 48 80 144
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for abl in "$@"; do
  touch $R/llm_quest_amd/csrc/attention.hip
  make -C $R/llm_quest_amd/csrc -j8 FLAGS_attention="-fno-slp-vectorize -DATTN_ABL=$abl" > /tmp/make_$abl.log 2>&1 || { tail -5 /tmp/make_$abl.log; exit 1; }
  echo "== ATTN_ABL=$abl"
  MI355_ATTN_DS_SPILL=${SPILL:-1} timeout -k 10 120 python3 $R/tools/attn_prof.py
done
