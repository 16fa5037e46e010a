#!/bin/bash
# r05: fused DOT query side -- parity test, timing against the separate launches; probes of the TransE query pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dot_query_side or query_pipeline" 2>&1 | tail -15 > gpurun_out/t2_tests.txt
{
for dbg in 0 1 2 4 8 16 32 63; do KGE_QP_DBG=$dbg KIND=transe timeout 120 python tools/qp_time.py 2>&1 | grep "query_pipeline\|Error"; done
KIND=transe timeout 120 python tools/qp_time.py 2>&1 | grep table_prep
KIND=complex N=40943 B=3134 timeout 120 python tools/qp_time.py 2>&1 | tail -2
KIND=distmult N=14951 B=29536 D=400 timeout 120 python tools/qp_time.py 2>&1 | tail -2
KIND=complex N=14541 B=20466 D=200 timeout 120 python tools/qp_time.py 2>&1 | tail -2
} > gpurun_out/t2_times.txt 2>&1
cat gpurun_out/t2_tests.txt gpurun_out/t2_times.txt
