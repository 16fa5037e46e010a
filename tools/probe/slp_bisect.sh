#!/bin/bash
# (r06) instruction-level bisection of the rare projection-epilogue miscount of lp_hi_stream.hip under the SLP vectoriser:
# variants of libkge_hip.so that differ ONLY in patched instructions of lp_hi_stream's assembly (tools/probe/asm_patch_build.py),
# each run three times through tools/dbg_pm.py (mismatching queries against the exact fp32 counts; 0 = correct).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
LIB=torchkge_amd/csrc/libkge_hip.so
cp $LIB /tmp/libkge_keep.so
for v in "$@"; do
  echo "== $v"
  cp tools/_libs/libkge_$v.so $LIB
  for i in 1 2 3; do timeout 300 python tools/dbg_pm.py 2>&1 | grep "mismatching" | sed 's/mismatching queries//' ; done
done
cp /tmp/libkge_keep.so $LIB
