#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/hs6
mkdir -p $OUT
export PYTHONUNBUFFERED=1
python tools/dbg_transh.py 2>&1 | grep "differing" 
python tools/dbg_transh.py transd_fb15k237 2>&1 | grep "differing"
timeout 1700 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
