set -e
B="python bench.py --steps 8 --warmup 2 --cpu-baseline off --optimizer off"
run() { MI355_LIB_PATH=$1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d.get('board_power',{}).get('mean_W'), d.get('board_power',{}).get('sclk_mean_MHz'))"; }
run llm_quest_amd/libmi355vlm.so base
run build_variants/libmi355vlm_caux16.so sc1
run build_variants/libmi355vlm_caux2.so nt
run build_variants/libmi355vlm_caux18.so sc1nt
run llm_quest_amd/libmi355vlm.so base
run build_variants/libmi355vlm_caux16.so sc1
