# Do stall cycles cost time under the power cap?  The persistent NT kernel with s_sleep N at the head of every K-tile (build_variants/libmi355vlm_sleepN.so:
# bash tools/build_variant.sh sleepN gemm_p2 "-DGEMM_SLEEP=N" with the s_sleep line at the top of kstep()), alone (random and all-zero operands: board power, clock, us) and in the step.
for v in base sleep2 sleep4 sleep8; do
  if [ $v = base ]; then lib=llm_quest_amd/libmi355vlm.so; else lib=build_variants/libmi355vlm_$v.so; fi
  echo "== $v"
  MI355_LIB_PATH=$lib python3 tools/power_trace.py gpurun_out/stall_$v.json nt nt_zero 2>&1 | grep -E "^nt"
  MI355_LIB_PATH=$lib python bench.py --steps 6 --warmup 2 --cpu-baseline off --optimizer off --pipe-probe off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['board_power']['mean_W'], d['board_power']['sclk_mean_MHz'])"
done
