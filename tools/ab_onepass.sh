mkdir -p gpurun_out/r06
(
bash tools/ab_env.sh "--workload complex_wn18rr" KGE_DOT_PREP_ONE_PASS=1 KGE_DOT_PREP_ONE_PASS=0
bash tools/ab_env.sh "--workload distmult_fb15k" KGE_DOT_PREP_ONE_PASS=1 KGE_DOT_PREP_ONE_PASS=0
export KGE_BENCH_TABLE_CACHE=/tmp/kge_cache
for v in 1 0 1 0; do
  echo "cfg5 KGE_DOT_PREP_ONE_PASS=$v: $(KGE_DOT_PREP_ONE_PASS=$v python bench.py --only-timed --workload complex_wikidata5m --steps 8 --warmup 3 --batch 8192 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done
) 2>&1 | grep -v amdgpu > gpurun_out/r06/dot_one_pass_ab.txt
cat gpurun_out/r06/dot_one_pass_ab.txt
