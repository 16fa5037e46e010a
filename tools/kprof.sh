#!/bin/bash
# per-kernel time table of a command (run on the GPU box):  bash tools/kprof.sh [args for bench.py]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kprof -o k -- python $R/bench.py --no-cpu-baseline "$@" > /tmp/kprof.log 2>&1
tail -1 /tmp/kprof.log | cut -c1-200
f=$(find /tmp/kprof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), r["Percentage"].rjust(7))
PY
