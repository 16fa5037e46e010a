#!/bin/bash
# PMC passes over the split count kernel (run on the GPU box): bash tools/split_pmc.sh "<counters pass 1>" "<pass 2>" ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/split_pmc
mkdir -p $OUT; rm -f $OUT/summary.txt
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc$i -o p -- python $R/tools/split_time.py > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY' >> $OUT/summary.txt
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k=r['Kernel_Name'][:48]
        if 'split_count' in k or 'lp_hi_stream' in k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
except Exception as e:
    print('ERR',e)
for k,v in agg.items():
    for c,vals in v.items():
        print('%-50s %-32s n=%d mean=%.5g'%(k,c,len(vals),sum(vals)/len(vals)))
PY
done
cat $OUT/summary.txt
