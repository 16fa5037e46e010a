mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --tb=short --no-header -p no:cacheprovider 2>&1 | tail -6
for extra in "" "--no-graph"; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline $extra 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$extra', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['filtered_mrr'])"; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload complex_wn18rr 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('complex', j['value'], j['ms_per_step'], j['roofline']['achieved'])"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload distmult_fb15k 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('distmult', j['value'], j['ms_per_step'], j['roofline']['achieved'])"
) > gpurun_out/run10.log 2>&1
cat gpurun_out/run10.log
