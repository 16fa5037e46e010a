#!/bin/bash
# Collect the evidence committed under profiles/ (run on the GPU box through gpurun):
#   bash tools/profile_round.sh r03
# kernel-trace/stats and PMC counters are collected in SEPARATE rocprofv3 runs.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-full-parity --no-traffic"
P="--no-cpu-baseline --no-traffic"          # (full-split parity ON: every workload's parity_full_split is recorded)
python $R/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_transe_fb15k237.json
python $R/bench.py --steps 20 --warmup 5 --kg uniform --weights xavier $Q 2>/dev/null | tail -1 > $OUT/bench_transe_fb15k237_uniformkg_xavier.json
python $R/bench.py --steps 10 --warmup 3 --workload complex_wn18rr $P 2>/dev/null | tail -1 > $OUT/bench_complex_wn18rr.json
python $R/bench.py --steps 5 --warmup 2 --workload distmult_fb15k $P 2>/dev/null | tail -1 > $OUT/bench_distmult_fb15k.json
python $R/bench.py --steps 10 --warmup 3 --no-split --no-cpu-baseline --no-full-parity 2>/dev/null | tail -1 > $OUT/bench_transe_fb15k237_nosplit.json
python $R/bench.py --steps 5 --warmup 2 --workload transh_fb15k237 $P 2>/dev/null | tail -1 > $OUT/bench_transh_fb15k237.json
python $R/bench.py --steps 5 --warmup 2 --workload transd_fb15k237 $P 2>/dev/null | tail -1 > $OUT/bench_transd_fb15k237.json
python $R/bench.py --steps 5 --warmup 2 --workload transe_l1_fb15k237 $P 2>/dev/null | tail -1 > $OUT/bench_transe_l1_fb15k237.json
python $R/bench.py --steps 10 --warmup 3 --l2-mode direct $Q 2>/dev/null | tail -1 > $OUT/bench_transe_fb15k237_l2direct.json
python $R/bench.py --steps 10 --warmup 3 --materialize $Q 2>/dev/null | tail -1 > $OUT/bench_transe_fb15k237_materialized.json
KGE_DEDUPE_QUERIES=0 python $R/bench.py --steps 20 --warmup 5 $Q --no-secondary 2>/dev/null | tail -1 > $OUT/bench_transe_fb15k237_no_query_columns.json
# cfg5 (ComplEx d=512, Wikidata5M shape, 18.8 GB of tables on one GPU): subsample parity + CPU sample inside
timeout 1200 python $R/bench.py --workload complex_wikidata5m --no-secondary --no-traffic --batch 8192 --kg uniform --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_complex_wikidata5m.json
# tiled top-k inference at cfg2 and cfg5 shapes
python $R/tools/topk_time.py --cfg5 2>/dev/null | grep "^{" > $OUT/topk_inference.jsonl
# per-kernel time of the bench command (the trained-weights set-up shows up as the score_fwd/bwd, key_* and optimiser rows)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 20 --warmup 5 $Q > $OUT/trace.log 2>&1
# ... and of the timed loop alone (no set-up training, no roofline / f32 legs): what one evaluate() consists of
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_eval -o bench -- python $R/bench.py --steps 20 --warmup 5 --only-timed --weights xavier > $OUT/trace_eval.log 2>&1
# ... and of the TransE-L1 evaluation (SAD prefilter)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_l1 -o bench -- python $R/bench.py --steps 10 --warmup 3 --only-timed --workload transe_l1_fb15k237 --weights xavier > $OUT/trace_l1.log 2>&1
# power / clock of the dominant kernel running back to back
bash $R/tools/power_probe.sh 0 2>&1 | grep -E "Power|sclk|launches" > $OUT/power_probe.txt
cd /tmp
# HBM-side counters of the same command, one --pmc pass per run (Xavier weights: no 500-step set-up under the counters)
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$tag -o bench -- python $R/bench.py --steps 3 --warmup 1 $Q --no-secondary --weights xavier --settle-ms 0 --no-graph > $OUT/pmc_$tag.log 2>&1
done
# ... and of the broadcast-subtract kernel (packed-FMA L2)
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmcdirect_SQ -o bench -- python $R/bench.py --steps 2 --warmup 1 $Q --no-secondary --weights xavier --settle-ms 0 --no-graph --l2-mode direct > $OUT/pmcdirect.log 2>&1
cd $R
python tools/summarize_profiles.py $OUT > $OUT/SUMMARY.md 2>&1
head -30 $OUT/SUMMARY.md
