export KGE_BENCH_TABLE_CACHE=/tmp/kge_cache
W="--workload distmult_fb15k"
python bench.py --only-timed --steps 30 --warmup 5 $W > /dev/null 2>&1
for rep in 1 2; do
for cfg in "KGE_REGION_SEGMENTS=0" "KGE_REGION_SEGMENTS=1" "KGE_RECHECK_REGION_WAVES=4" "KGE_RECHECK_REGION_WAVES=4 KGE_REGION_MAX_BYTES=20480" "KGE_RECHECK_REGION_WAVES=2 KGE_REGION_MAX_BYTES=20480" "KGE_RECHECK_REGION_WAVES=4 KGE_REGION_MAX_BYTES=12288"; do
  echo "$cfg: $(env $cfg python bench.py --only-timed --steps 200 --warmup 20 $W 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done; done
