mkdir -p gpurun_out/r06
export KGE_BENCH_TABLE_CACHE=/tmp/kge_cache
(
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsplit.py -x -q -m gpu -k "transh or transd or proj" 2>&1 | tail -3
for rep in 1 2; do for v in 1 0; do for w in transh_fb15k237 transd_fb15k237; do
  echo "KGE_PREP_SIDE_STREAM=$v $w: $(KGE_PREP_SIDE_STREAM=$v python bench.py --only-timed --steps 200 --warmup 20 --workload $w 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done; done; done
bash tools/eval_timeline.sh gpurun_out/r06/timeline_transh_side.txt --workload transh_fb15k237 > /dev/null 2>&1
cat gpurun_out/r06/timeline_transh_side.txt
) > gpurun_out/r06/prep_side_stream_ab.txt 2>&1
cat gpurun_out/r06/prep_side_stream_ab.txt
