#!/bin/bash
# PMC passes over the split count kernel (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/split_pmc
mkdir -p $OUT
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU" \
            "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc$i -o p -- python $R/tools/split_dev.py time > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY' >> $OUT/summary.txt
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k=r['Kernel_Name'][:60]
        if 'split_count' in k or 'recheck' in k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
except Exception as e:
    print('ERR',e)
for k,v in agg.items():
    for c,vals in v.items():
        print('%-62s %-32s n=%d mean=%.4g'%(k,c,len(vals),sum(vals)/len(vals)))
PY
done
cat $OUT/summary.txt
