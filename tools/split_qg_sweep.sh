#!/bin/bash
# work-order experiment of the split count kernel: query panels interleaved per candidate sweep (KGE_SPLIT_QG)
# -> kernel time (rocprofv3 kernel-trace) and L2-miss traffic (FETCH_SIZE, separate pass) per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for qg in "$@"; do
  export KGE_SPLIT_QG=$qg
  rm -rf /tmp/qg_t /tmp/qg_p
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qg_t -o b -- python $R/bench.py --steps 30 --warmup 5 --only-timed --weights xavier > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/qg_p -o b -- python $R/bench.py --steps 3 --warmup 0 --only-timed --no-graph --weights xavier --settle-ms 0 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
st = glob.glob('/tmp/qg_t/**/b_kernel_stats.csv', recursive=True)[0]
t = {r['Name']: float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(st)) if 'lp_split_count' in r['Name']}
f = collections.defaultdict(list)
for fn in glob.glob('/tmp/qg_p/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if r['Counter_Name'] == 'FETCH_SIZE' and 'lp_split_count' in r['Kernel_Name']:
            f[r['Kernel_Name']].append(float(r['Counter_Value']))
print('QG=$qg', 'kernel us:', {k[-22:]: round(v, 1) for k, v in t.items()}, 'sum', round(sum(t.values()), 1),
      '| FETCH_SIZE MB (x2 corrected):', {k[-22:]: round(2 * sum(v) / len(v) / 1024, 1) for k, v in f.items()},
      'sum', round(sum(2 * sum(v) / len(v) / 1024 for v in f.values()), 1))
PY
done
