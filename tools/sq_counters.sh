#!/bin/bash
# SQ instruction / activity counters of the path's kernels (rocprofv3 --pmc, kernel-trace only,
# eager launches): gpurun -- 'bash tools/sq_counters.sh'   -> gpurun_out/sqpmc/
# usage: bash tools/sq_counters.sh [frames per launch, default 1]
R=$(pwd)
B=${1:-1}
mkdir -p "$R/gpurun_out/sqpmc"
cd /tmp; export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $grp | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$R/gpurun_out/sqpmc/$n" -- python3 "$R/bench.py" --batch $B --main-only --min-window-s 0.01 --warmup-s 0.01 --steps 3 --warmup 1 --no-cpu-baseline --no-batched --no-graph > "$R/gpurun_out/sqpmc/$n.log" 2>&1
done
find "$R/gpurun_out/sqpmc" -name "*.db" -delete
find "$R/gpurun_out/sqpmc" -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob, collections, re, os
root = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'sqpmc')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/*/*/*counter_collection.csv') + glob.glob(root + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        m = re.search(r'chain_kernel<(\d+), *(\d+)', k)
        k2 = 'chain_dual' if 'chain_dual' in k else ('chain<%s,%s>' % m.groups() if m else ('self_attn' if 'self_attn' in k else
              ('box_decode' if 'box_decode' in k else ('radar_compact' if 'radar_hit' in k or 'radar_part' in k else None))))
        if k2:
            acc[k2][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print('   %-32s %16.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
