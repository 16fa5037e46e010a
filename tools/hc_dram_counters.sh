cd /tmp; export TMPDIR=/tmp
for qg in 32 8; do
  export KGE_HS_QG=$qg; rm -rf /tmp/hcpmc
  VARIANTS=hc4 REPS=1 timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace --output-format csv -d /tmp/hcpmc -o t -- python $GRAFT_REPO_ROOT/tools/hc_time.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
v = collections.defaultdict(list)
for f in glob.glob('/tmp/hcpmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lp_hi_chunk_kernel' in r.get('Kernel_Name', ''):
            v[r['Counter_Name']].append(float(r['Counter_Value']))
print('QG=$qg', {k: '%.3e' % (sum(x)/len(x)) for k, x in v.items()})
PY
done
