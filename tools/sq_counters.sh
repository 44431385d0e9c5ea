#!/bin/bash
# SQ instruction / activity counters of the path's kernels (rocprofv3 --pmc, kernel-trace only,
# eager launches): gpurun -- 'bash tools/sq_counters.sh'   -> gpurun_out/sqpmc/
R=$(pwd)
mkdir -p "$R/gpurun_out/sqpmc"
cd /tmp; export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $grp | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$R/gpurun_out/sqpmc/$n" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-batched --no-graph > "$R/gpurun_out/sqpmc/$n.log" 2>&1
done
find "$R/gpurun_out/sqpmc" -name "*.db" -delete
find "$R/gpurun_out/sqpmc" -name "*kernel_trace.csv" -delete
ls -R "$R/gpurun_out/sqpmc" | head -20
