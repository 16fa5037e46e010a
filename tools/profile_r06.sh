#!/bin/bash
# r06 evidence (one box, one run): bench lines of every workload incl. cfg5 (trained-like, chunked-panel kernel), the dispatch
# timeline and the evaluate-only kernel statistics of each (trained tables come from KGE_BENCH_TABLE_CACHE, so the traces hold
# NO training kernels), counters of the chunked-panel kernel at cfg5's shape.   bash tools/profile_r06.sh
TAG=r06
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export KGE_BENCH_TABLE_CACHE=/tmp/kge_cache
T="--no-cpu-baseline"
b() { name=$1; shift; timeout 1500 python $R/bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/bench_$name.json; [ -s $OUT/bench_$name.json ] && rm -f $OUT/$name.err; }
tl() { name=$1; shift; cd $R; timeout 600 bash tools/eval_timeline.sh gpurun_out/profiles_$TAG/timeline_$name.txt "$@" > /dev/null 2>&1; cd /tmp; }
ks() {  # evaluate-only kernel statistics: the bench's timed loop under rocprofv3 --kernel-trace --stats, tables from the cache
  name=$1; shift; rm -rf /tmp/ks_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$name -o bench -- python $R/bench.py --only-timed "$@" > $OUT/trace_$name.log 2>&1
  f=$(find /tmp/ks_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_evaluate_only_$name.csv
  rm -rf /tmp/ks_$name; }
# the headline line exactly as the driver runs it, then the other BASELINE workloads
b transe_fb15k237 --steps 20 --warmup 3
b transe_fb15k237_three_products --steps 20 --warmup 3 --split-level 0 $T --no-secondary
b complex_wn18rr --steps 10 --warmup 3 --workload complex_wn18rr $T
b distmult_fb15k --steps 5 --warmup 2 --workload distmult_fb15k $T
b transh_fb15k237 --steps 20 --warmup 3 --workload transh_fb15k237 $T
b transd_fb15k237 --steps 20 --warmup 3 --workload transd_fb15k237 $T
b complex_wikidata5m --workload complex_wikidata5m --no-secondary --steps 5 --warmup 2 --batch 8192
# one steady-state evaluate() dispatch by dispatch
tl transe_fb15k237
tl complex_wn18rr --workload complex_wn18rr
tl distmult_fb15k --workload distmult_fb15k
tl transh_fb15k237 --workload transh_fb15k237
tl transd_fb15k237 --workload transd_fb15k237
tl complex_wikidata5m --workload complex_wikidata5m --no-secondary --batch 8192 --steps 4
# kernel statistics of the timed loop alone
ks transe_fb15k237 --steps 20 --warmup 5
ks complex_wn18rr --steps 10 --warmup 3 --workload complex_wn18rr
ks distmult_fb15k --steps 5 --warmup 2 --workload distmult_fb15k
ks transh_fb15k237 --steps 10 --warmup 3 --workload transh_fb15k237
ks transd_fb15k237 --steps 10 --warmup 3 --workload transd_fb15k237
ks complex_wikidata5m --steps 4 --warmup 1 --workload complex_wikidata5m --no-secondary --batch 8192
# counters of the chunked-panel kernel at cfg5's shape (kernel alone, tools/hc_time.py; one --pmc group per pass)
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1); rm -rf /tmp/hcp
  VARIANTS=hc4 REPS=1 timeout 400 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/hcp -o t -- python $R/tools/hc_time.py > /dev/null 2>&1
  python3 - "$pass" >> $OUT/hi_chunk_counters.txt <<'PY'
import csv, glob, sys, collections
v = collections.defaultdict(list)
for f in glob.glob('/tmp/hcp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lp_hi_chunk_kernel' in r.get('Kernel_Name', ''):
            v[r['Counter_Name']].append(float(r['Counter_Value']))
print('pass [%s]: ' % sys.argv[1] + ', '.join('%s = %.4g' % (k, sum(x) / len(x)) for k, x in sorted(v.items())) + '   (per launch, lp_hi_chunk_kernel<8,4,65,13>, 10,266 queries x 4,594,485 candidates)')
PY
done
cd $R
timeout 300 python tools/first_call.py 2>/dev/null | tail -1 > $OUT/first_call.json
timeout 600 python tools/topk_time.py 2>/dev/null | grep "^{" > $OUT/topk_inference.jsonl
python tools/summarize_profiles.py $OUT > $OUT/SUMMARY.md 2>&1
du -sh $OUT; ls $OUT | head -80
