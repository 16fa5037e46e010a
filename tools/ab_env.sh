# A/B of environment switches on one box: bash tools/ab_env.sh "<workload args>" VAR=a VAR=b ...   (alternating, 3 rounds)
W="$1"; shift
export KGE_BENCH_TABLE_CACHE=/tmp/kge_cache
python bench.py --only-timed --steps 50 --warmup 5 $W > /dev/null 2>&1
for rep in 1 2 3; do for kv in "$@"; do
  echo "$kv: $(env $kv python bench.py --only-timed --steps 300 --warmup 30 $W 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done; done
