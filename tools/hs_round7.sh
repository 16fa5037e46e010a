#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/hs7
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 1700 python -m pytest tests -x -q -m gpu --deselect "tests/test_gpu_fullsplit.py::test_full_test_split_vs_gpu_resident_reference" -k "not one_product and not projection_modes and not table_prep" > $OUT/pytest_gpu.log 2>&1
tail -12 $OUT/pytest_gpu.log
