#!/bin/bash
# Round profile on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r1'
# Writes under gpurun_out/<tag>/ : GPU test log, bench JSON line, rocprofv3
# kernel stats + trace of the same bench command, and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only).  Copy what is to
# be judged into profiles/ with tools/collect_profile.py.
set -u
TAG=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py --batch 2 --no-cpu-baseline --no-batched > "$OUT/bench_b2.json" 2>> "$OUT/bench.err"
python bench.py --batch 4 --no-cpu-baseline --no-batched > "$OUT/bench_b4.json" 2>> "$OUT/bench.err"
python bench.py --train --steps 50 --warmup 5 > "$OUT/bench_train.json" 2>> "$OUT/bench.err"
cd /tmp
# per-kernel durations: one frame at a time (with frames in flight the kernels of different frames share the GPU)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$REPO/bench.py" --lanes 1 --steps 50 --warmup 5 --no-cpu-baseline --no-batched > "$OUT/prof.log" 2>&1
# the default command (3 frames in flight): trace of the overlap
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_lanes" -- python3 "$REPO/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --no-batched > "$OUT/prof_lanes.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-batched --no-graph > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-batched --no-graph > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/pmc_mfma" -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-batched --no-graph > "$OUT/pmc_mfma.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_train" -- python3 "$REPO/bench.py" --train --steps 20 --warmup 3 > "$OUT/prof_train.log" 2>&1
cd "$REPO"
# keep the merge-back small: stats csv + counter csv only
find "$OUT" -name '*.db' -delete 2>/dev/null
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete 2>/dev/null
ls -R "$OUT" | head -50
tail -3 "$OUT/pytest_gpu.log"; cat "$OUT/bench.json"
