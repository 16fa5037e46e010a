#!/bin/bash
# round-4 GPU call 1: GPU test suite, N>1 dry runs (gloo x2 on one GPU, one forced RCCL rank), cfg4 / cfg5 in their
# sharded form, band-occupancy probe of a one-product f16 sweep
cd ${GRAFT_REPO_ROOT:-.}
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests1.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests1.log
tail -3 $O/tests1.log
timeout 900 bash tools/dry_subset.sh > $O/n2_dryrun_gloo.txt 2>&1
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
timeout 600 $TR --master-port 29541 tools/sharded_shapes.py --workload distmult_fb15k --out $O/sharded_distmult_fb15k.json > $O/sharded_cfg4.log 2>&1; echo "cfg4 rc=$?"
timeout 900 $TR --master-port 29542 tools/sharded_shapes.py --workload complex_wikidata5m --batch 8192 --score-facts 64 --out $O/sharded_complex_wikidata5m.json > $O/sharded_cfg5.log 2>&1; echo "cfg5 rc=$?"
for w in trained xavier; do timeout 300 python tools/band_probe.py --weights $w 2>/dev/null | tail -1 >> $O/band_probe.jsonl; done
timeout 300 python tools/band_probe.py --workload distmult_fb15k --weights trained 2>/dev/null | tail -1 >> $O/band_probe.jsonl
cat $O/band_probe.jsonl; tail -2 $O/sharded_cfg4.log $O/sharded_cfg5.log; cat $O/n2_dryrun_gloo.txt | cut -c1-1500
