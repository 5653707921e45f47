B="python bench.py --steps 8 --warmup 2 --cpu-baseline off --optimizer off --pipe-probe off"
for v in 0 1 0 1; do MI355_ABLATE_QKNORM=$v $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate', $v, d['ms_per_step'], d['board_power']['mean_W'], d['loss'])"; done
