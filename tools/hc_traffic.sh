#!/bin/bash
# (r06) chunked-panel count kernel at cfg5 shape: time and L2-miss traffic (FETCH_SIZE pass) per work order
#   bash tools/hc_traffic.sh OUT  [QG values ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
: > $OUT
for qg in "$@"; do
  for nt in 4 3; do
    export KGE_HS_QG=$qg
    echo "== KGE_HS_QG=$qg NT=$nt" >> $OUT
    VARIANTS=hc$nt REPS=3 timeout 300 python $R/tools/hc_time.py 2>&1 | grep "^hc" >> $OUT
    rm -rf /tmp/hcpmc
    VARIANTS=hc$nt REPS=1 timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/hcpmc -o t -- python $R/tools/hc_time.py > /dev/null 2>&1
    python3 - >> $OUT <<'PY'
import csv, glob
v = []
for f in glob.glob('/tmp/hcpmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') == 'FETCH_SIZE' and 'lp_hi_chunk_kernel' in r.get('Kernel_Name', ''):
            v.append(float(r['Counter_Value']))
if v:
    print('   FETCH_SIZE per launch: %.1f GB reported, x2 (gfx950 wide-read correction) = %.1f GB  (%d launches)' % (sum(v) / len(v) * 1024 / 1e9, 2 * sum(v) / len(v) * 1024 / 1e9, len(v)))
PY
  done
done
cat $OUT
