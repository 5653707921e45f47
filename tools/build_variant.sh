#!/bin/bash
# Profiling builds: one copy of libmi355vlm.so with extra -D flags on ONE source, kept under build_variants/ (git-ignored, travels to the GPU box).
#   bash tools/build_variant.sh <tag> <source stem: attention | gemm_p2 | ...> "<extra flags>"
# Use on the box through MI355_LIB_PATH=build_variants/libmi355vlm_<tag>.so (llm_quest_amd/_lib.py).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/llm_quest_amd/csrc; tag=$1; stem=$2; extra=$3
mkdir -p $R/build_variants/obj
base="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffp-contract=off"
case $stem in
  attention) src=attention.hip; flags="-fno-slp-vectorize";;
  gemm_p*) src=gemm.hip; flags="-DGEMM_PART=${stem#gemm_p}";;
  *) src=$stem.hip; flags="";;
esac
/opt/rocm/bin/hipcc $base $flags $extra -c $C/$src -o $R/build_variants/obj/${stem}_$tag.o
objs=""
for o in $C/*.o; do b=$(basename $o .o); [ "$b" = "$stem" ] && objs="$objs $R/build_variants/obj/${stem}_$tag.o" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/libmi355vlm_$tag.so $objs
echo built build_variants/libmi355vlm_$tag.so
