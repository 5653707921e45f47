for g in base gm2 gm3 gm4 gm8 gm12 gm16 base; do
  if [ $g = base ]; then lib=llm_quest_amd/libmi355vlm.so; else lib=build_variants/libmi355vlm_$g.so; fi
  echo "== $g"; MI355_LIB_PATH=$lib python tools/experimental/ab_group_m.py
done
