# the K-step burst of bench.py against the bracket period and the warm-up length:  bash profiles/burst_vs_warmup.sh
for cfg in "0 5 20" "0 5 20" "7 5 20" "7 5 20" "21 5 20" "21 5 20" "0 3000 20" "7 3000 20" "0 5 200" "7 5 200"; do set -- $cfg
TBK_PROF_PERIOD=$1 python bench.py --steps $3 --warmup $2 --headline-only --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('period $1 warmup $2 steps $3', d['ms_per_step'], d['kernels'])"
done
