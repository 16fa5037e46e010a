#!/bin/bash
# Collect the evidence committed under profiles/ (run on the GPU box through gpurun):
#   bash tools/profile_round.sh r04
# kernel-trace/stats and PMC counters are collected in SEPARATE rocprofv3 runs.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-full-parity --no-traffic"
P="--no-cpu-baseline --no-traffic"          # (full-split parity ON: every workload's parity_full_split is recorded)
T="--no-cpu-baseline"                       # (... and the run's own PMC passes: traffic, SQ_INSTS_MFMA, MFMA busy)
b() { name=$1; shift; timeout 600 python $R/bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/bench_$name.json; [ -s $OUT/bench_$name.json ] && rm -f $OUT/$name.err; }
tr() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o bench -- python $R/bench.py --only-timed "$@" > $OUT/trace_$name.log 2>&1; }
finish() {
  cd $R
  python tools/summarize_profiles.py $OUT > $OUT/SUMMARY.md 2>&1
  # keep what gets committed small: the kernel-stats / counter CSVs and logs, not the raw traces
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
  du -sh $OUT; head -30 $OUT/SUMMARY.md | cut -c1-400
}
# bash tools/profile_round.sh r04 <workload> ...: refresh the bench line + the evaluate-only kernel trace of the named
# workloads only (after a change that touches just them); SUMMARY.md is then rebuilt here from the merged directory
if [ $# -gt 1 ]; then
  shift
  for w in "$@"; do
    case $w in transh_fb15k237) t=transh;; transd_fb15k237) t=transd;; transe_fb15k237) t=eval;; *) t=$w;; esac   # (trace names of the full run)
    b $w --steps 10 --warmup 3 --workload $w $T; tr $t --steps 10 --warmup 3 --workload $w
  done
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
  exit 0
fi
# the headline line exactly as the driver runs it (all legs on), then the variants
b transe_fb15k237 --steps 20 --warmup 3
b transe_fb15k237_three_products --steps 20 --warmup 3 --split-level 0 $T --no-secondary
b transe_fb15k237_uniformkg_xavier --steps 20 --warmup 5 --kg uniform --weights xavier $Q
b complex_wn18rr --steps 10 --warmup 3 --workload complex_wn18rr $T
b distmult_fb15k --steps 5 --warmup 2 --workload distmult_fb15k $T
b transe_fb15k237_nosplit --steps 10 --warmup 3 --no-split --no-cpu-baseline --no-full-parity
b transh_fb15k237 --steps 5 --warmup 2 --workload transh_fb15k237 $T
b transd_fb15k237 --steps 5 --warmup 2 --workload transd_fb15k237 $T
b transe_l1_fb15k237 --steps 5 --warmup 2 --workload transe_l1_fb15k237 $P
b transe_fb15k237_l2direct --steps 10 --warmup 3 --l2-mode direct $Q
b transe_fb15k237_materialized --steps 10 --warmup 3 --materialize $Q
KGE_DEDUPE_QUERIES=0 timeout 400 python $R/bench.py --steps 20 --warmup 5 $Q --no-secondary 2>/dev/null | tail -1 > $OUT/bench_transe_fb15k237_no_query_columns.json
# cfg5 (ComplEx d=512, Wikidata5M shape, 18.8 GB of tables on one GPU): subsample parity + CPU sample + traffic inside
timeout 1500 python $R/bench.py --workload complex_wikidata5m --no-secondary --batch 8192 --kg uniform --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_complex_wikidata5m.json
# tiled top-k inference at cfg2 and cfg5 shapes; the first evaluate() of a process
timeout 600 python $R/tools/topk_time.py --cfg5 2>/dev/null | grep "^{" > $OUT/topk_inference.jsonl
timeout 300 python $R/tools/first_call.py 2>/dev/null | tail -1 > $OUT/first_call.json
# per-kernel time of the bench command (the trained-weights set-up shows up as the score_fwd/bwd, key_* and optimiser rows)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 20 --warmup 5 $Q > $OUT/trace.log 2>&1
# ... and of the timed loops alone (no roofline / f32 legs): what one evaluate() consists of, per workload
tr eval --steps 20 --warmup 5
tr eval_three_products --steps 20 --warmup 5 --split-level 0
tr l1 --steps 10 --warmup 3 --workload transe_l1_fb15k237 --weights xavier
tr l2direct --steps 10 --warmup 3 --l2-mode direct --weights xavier
tr transd --steps 10 --warmup 3 --workload transd_fb15k237
tr transh --steps 10 --warmup 3 --workload transh_fb15k237
tr complex_wn18rr --steps 10 --warmup 3 --workload complex_wn18rr
tr distmult_fb15k --steps 5 --warmup 2 --workload distmult_fb15k
tr complex_wikidata5m --steps 2 --warmup 1 --workload complex_wikidata5m --batch 8192 --kg uniform
# power / clock of the dominant kernel running back to back
bash $R/tools/power_probe.sh 0 2>&1 | grep -E "Power|sclk|launches" > $OUT/power_probe.txt
cd /tmp
# HBM-side and matrix-pipe counters of the evaluate() kernels, one --pmc pass per run: the one-product level (forced on
# Xavier weights: same kernel work, no 500-step set-up under the counters) and the three-product sweep
for lv in 1 0; do
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_level${lv}_$tag -o bench -- python $R/bench.py --only-timed --steps 3 --warmup 1 --weights xavier --settle-ms 0 --no-graph --split-level $lv > $OUT/pmc_level${lv}_$tag.log 2>&1
done
done
# ... of the exact fp32 MFMA counts (--no-split: the overflow fallback / score-matrix kernel) ...
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_nosplit_$pass -o bench -- python $R/bench.py --only-timed --steps 3 --warmup 1 --weights xavier --settle-ms 0 --no-graph --no-split > $OUT/pmc_nosplit_$pass.log 2>&1
done
# ... and of the broadcast-subtract kernel (packed-FMA L2)
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmcdirect_SQ -o bench -- python $R/bench.py --only-timed --steps 2 --warmup 1 --weights xavier --settle-ms 0 --no-graph --l2-mode direct > $OUT/pmcdirect.log 2>&1
finish
