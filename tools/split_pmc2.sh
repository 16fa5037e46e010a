#!/bin/bash
# Where do the count kernel's waves wait?  Texture-addresser / LDS / VMEM counters of lp_split_count_kernel, one --pmc
# pass per group (no trace domains besides the kernel trace):   bash tools/split_pmc2.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
  "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
  "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
  "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU" \
  "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" \
  "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY TCC_BUSY_sum TCC_CYCLE_sum" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/p$i -o b -- python $R/bench.py --steps 3 --warmup 1 --only-timed --no-secondary --weights xavier --settle-ms 0 > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY'
import collections, csv, glob, os, sys
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], 'p*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lp_split_count' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    v = agg[k]
    print('%-40s launches %3d  mean per launch %.6g' % (k, len(v), sum(v) / len(v)))
PY
