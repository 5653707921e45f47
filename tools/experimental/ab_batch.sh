for b in 160 80 40 24 160; do
python bench.py --batch $b --steps 8 --warmup 2 --cpu-baseline off --optimizer off --pipe-probe off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', $b, d['ms_per_step'], d['roofline']['frac'], d['board_power']['mean_W'], d['board_power']['sclk_mean_MHz'])"
done
