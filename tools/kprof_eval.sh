#!/bin/bash
# per-kernel time table of the evaluate() kernels only (trained default workload):  bash tools/kprof_eval.sh [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kprof -o k -- python $R/bench.py --only-timed "$@" > /tmp/kprof.log 2>&1
tail -1 /tmp/kprof.log | cut -c1-160
f=$(find /tmp/kprof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keys = ('lp_split', 'hi_rows', 'fill', 'lp_l1_sad', 'sad_', 'lp_direct', 'recheck', 'query_pipeline', 'fsub', 'split_rows', 'row_sqnorm', 'prefix_max', 'rank_finalize', 'copyBuffer', 'lp_gemm', 'split_thr', 'pair_scores', 'lp_prep', 'absmax')
for r in rows:
    if any(k in r["Name"] for k in keys):
        print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9))
PY
