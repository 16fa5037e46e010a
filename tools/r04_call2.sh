#!/bin/bash
# round-4 GPU call 2: GPU test suite, one-product probes of the count kernel, first-call timing, forced-RCCL-rank dry run
cd ${GRAFT_REPO_ROOT:-.}
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/tests2.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests2.log
tail -5 $O/tests2.log
for dbg in 0 1024 3072 2048 0 1024 3072; do KGE_SPLIT_DBG=$dbg timeout 120 python tools/split_time.py 2>&1 | grep count; done | tee $O/split_probe_one_product.txt
timeout 300 python tools/first_call.py 2>/dev/null | tail -1 | tee $O/first_call.json
export KGE_FORCE_COLLECTIVES=1
nrun() { timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-weak "$@" 2>/dev/null | tail -1 | grep -o '"value": [0-9.e+]*\|"ms_per_step": [0-9.]*\|"parallelism": "[^"]*"\|"collective_time": {[^}]*}\|"other_exchange": {[^}]*}[^}]*}' | paste -s -d' '; }
echo "== 1 RCCL rank, collectives forced: default (scores all-to-all headline, counts beside it)" | tee $O/n1_forced_rccl.txt
nrun | tee -a $O/n1_forced_rccl.txt
