#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kprof -o k -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-parity --weights xavier > /tmp/kprof.log 2>&1
f=$(find /tmp/kprof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = ('lp_split', 'recheck', 'query_pipeline', 'fsub', 'split_rows', 'row_sqnorm', 'prefix_max', 'rank_finalize', 'lp_gemm', 'split_thr', 'pair_scores', 'lp_prep', 'absmax')
for r in rows[:60]:
    if not any(k in r["Name"] for k in skip):
        print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), ("%.2f" % (float(r["TotalDurationNs"]) / 1e6)).rjust(9))
PY
