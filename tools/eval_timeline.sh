#!/bin/bash
# (r05) timeline of ONE steady-state evaluate() -- every dispatch of the hipGraph replay with its start offset, duration and
# the gap to the end of the previous dispatch:  bash tools/eval_timeline.sh OUTFILE [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktl
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktl -o k -- python $R/bench.py --only-timed --steps 30 "$@" > /tmp/ktl.log 2>&1
tail -1 /tmp/ktl.log | cut -c1-200
f=$(find /tmp/ktl -name "*kernel_trace.csv" | head -1)
python3 - "$f" > "$R/$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name']
fin = [i for i, r in enumerate(rows) if 'rank_finalize' in name(r)]
# the last complete evaluate: from the dispatch after the second-to-last finalize to the last finalize
import os
back = int(os.environ.get('TL_BACK', '0'))       # TL_BACK=1: the evaluate (or batch) before the last one
a, b = fin[-2 - back] + 1, fin[-1 - back]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = None
print('# one steady-state evaluate(): %d dispatches, %.1f us from the first start to the last end' % (
    b - a + 1, (int(rows[b]['End_Timestamp']) - t0) / 1e3))
print('%-72s %9s %9s %8s  %s' % ('kernel', 'start_us', 'dur_us', 'gap_us', 'queue'))
busy = 0
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = '' if prev_end is None else '%.1f' % ((s - prev_end) / 1e3)
    print('%-72s %9.1f %9.1f %8s  %s' % (name(r)[:72], (s - t0) / 1e3, (e - s) / 1e3, gap, r.get('Queue_Id', '')))
    prev_end = e if prev_end is None else max(prev_end, e)
# steady-state period: finalize-to-finalize over the last 10 evaluates
per = [(int(rows[fin[i]]['End_Timestamp']) - int(rows[fin[i - 1]]['End_Timestamp'])) / 1e3 for i in range(len(fin) - 10, len(fin))]
print('# finalize-to-finalize period of the last 10 evaluates (us):', ' '.join('%.0f' % x for x in per))
PY
cat "$R/$OUT" | cut -c1-140
