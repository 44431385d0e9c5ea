#!/bin/bash
# SQ counters of the attention core on one large problem (B 2, Q 3600): gpurun -- 'bash tools/attn_pmc.sh'
R=$(pwd); O=$R/gpurun_out/attn_pmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_BRANCH"; do
  n=$(echo $grp | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$n -- python3 $R/tools/attn_scaling.py pmc > $O/$n.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/root/repo/gpurun_out/attn_pmc/*/*/*counter_collection.csv') + glob.glob('/root/repo/gpurun_out/attn_pmc/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'self_attn' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    v = acc[k]
    print('%-32s %16.0f  (n=%d)' % (k, sum(v) / len(v), len(v)))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
