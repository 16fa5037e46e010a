#!/bin/bash
# r05 evidence (trimmed profile_round.sh: ~15 GPU-minutes): bench lines, evaluate timelines, kernel stats, PMC passes of the
# free-running one-product kernel, power probe, N > 1 dry runs.   bash tools/profile_r05.sh
TAG=r05
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T="--no-cpu-baseline"
b() { name=$1; shift; timeout 600 python $R/bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/bench_$name.json; [ -s $OUT/bench_$name.json ] && rm -f $OUT/$name.err; }
tl() { name=$1; shift; cd $R; timeout 400 bash tools/eval_timeline.sh gpurun_out/profiles_$TAG/timeline_$name.txt "$@" > /dev/null 2>&1; cd /tmp; }
tr() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o bench -- python $R/bench.py --only-timed "$@" > $OUT/trace_$name.log 2>&1; }
# the headline line exactly as the driver runs it, then the other BASELINE workloads and the A/B variants on the same box
b transe_fb15k237 --steps 20 --warmup 3
KGE_HI_STREAM=0 timeout 400 python $R/bench.py --steps 20 --warmup 3 $T --no-secondary --no-full-parity 2>/dev/null | tail -1 > $OUT/bench_transe_fb15k237_r04_level1_kernel.json
b transe_fb15k237_three_products --steps 20 --warmup 3 --split-level 0 $T --no-secondary
b complex_wn18rr --steps 10 --warmup 3 --workload complex_wn18rr $T
b distmult_fb15k --steps 5 --warmup 2 --workload distmult_fb15k $T
b transh_fb15k237 --steps 5 --warmup 2 --workload transh_fb15k237 $T
b transd_fb15k237 --steps 5 --warmup 2 --workload transd_fb15k237 $T
for w in complex_wn18rr distmult_fb15k transh_fb15k237; do
  KGE_HI_STREAM=0 timeout 400 python $R/bench.py --steps 5 --warmup 2 --workload $w $T --no-secondary --no-full-parity --no-traffic 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w r04 level-1 kernel (KGE_HI_STREAM=0), same box:', d['ms_per_step'], 'ms per evaluate; count kernel', d['roofline']['kernel_ms'], 'ms')" >> $OUT/level1_kernel_ab.txt
done
# one steady-state evaluate() dispatch by dispatch
tl transe_fb15k237
tl complex_wn18rr --workload complex_wn18rr
tl distmult_fb15k --workload distmult_fb15k
tl transh_fb15k237 --workload transh_fb15k237
tl transd_fb15k237 --workload transd_fb15k237
# kernel stats of the timed loop alone
tr eval --steps 20 --warmup 5
tr complex_wn18rr --steps 10 --warmup 3 --workload complex_wn18rr
tr distmult_fb15k --steps 5 --warmup 2 --workload distmult_fb15k
tr transh --steps 10 --warmup 3 --workload transh_fb15k237
tr transd --steps 10 --warmup 3 --workload transd_fb15k237
# counters of the evaluate() kernels, one --pmc pass per run (forced level 1 on Xavier weights: same kernel work)
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_level1_$tag -o bench -- python $R/bench.py --only-timed --steps 3 --warmup 1 --weights xavier --settle-ms 0 --no-graph --split-level 1 > $OUT/pmc_level1_$tag.log 2>&1
done
# power / clock of the free-running kernel back to back, and of its probes
cd $R
{ for pr in 0 16; do echo "== KGE_HS_PROBE=$pr"; LEVEL=1 TAIL=1 FRAG=1 KGE_HS_PROBE=$pr bash tools/power_probe.sh 0 2>&1 | grep -E "Power|sclk|launches"; done; } > $OUT/power_probe_hi_stream.txt
{ for pr in 0 1 2 4 8 12 16; do LEVEL=1 TAIL=1 FRAG=1 K=200 KGE_HS_PROBE=$pr timeout 120 python tools/split_time.py 2>&1 | grep count | sed "s/^/probe=$pr /"; done
  LEVEL=1 TAIL=1 FRAG=0 K=200 timeout 120 python tools/split_time.py 2>&1 | grep count | sed "s/^/r04 kernel /"
  for K in 400; do for f in 0 1; do LEVEL=1 TAIL=1 FRAG=$f K=$K timeout 120 python tools/split_time.py 2>&1 | grep count; done; done; } > $OUT/hi_stream_probes.txt
timeout 300 python tools/first_call.py 2>/dev/null | tail -1 > $OUT/first_call.json
timeout 600 python tools/topk_time.py 2>/dev/null | grep "^{" > $OUT/topk_inference.jsonl
# N > 1 logic on the one GPU: two gloo ranks, then one RCCL rank with the collectives forced
timeout 1500 bash tools/dry_subset.sh > $OUT/n2_dryrun_gloo.txt 2>&1
cd $R
python tools/summarize_profiles.py $OUT > $OUT/SUMMARY.md 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; ls $OUT | head -60
