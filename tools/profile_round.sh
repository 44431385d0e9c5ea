#!/bin/bash
# Round profile on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r2'
# Writes under gpurun_out/<tag>/ : GPU test log, bench JSON lines, rocprofv3 kernel stats + traces
# of the same bench command, and the PMC passes (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES:
# separate runs, kernel-trace only).  Copy what is to be judged into profiles/ with
# tools/collect_profile.py <tag> <name>.
set -u
TAG=${1:-r6}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ -z "${SKIP_TESTS:-}" ]; then
python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
fi
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver.json" 2> "$OUT/bench.err"      # the driver's command, everything in one line
python bench.py --shapes vovnet --no-configs --no-batched --no-handoff > "$OUT/bench_vovnet.json" 2>> "$OUT/bench.err"   # configs[4] shapes as the headline
python bench.py --matrix-path f32 --no-configs --no-batched --no-handoff --no-cpu-baseline > "$OUT/bench_f32.json" 2>> "$OUT/bench.err"   # the exact-f32 chains
python bench.py > "$OUT/bench.json" 2>> "$OUT/bench.err"                                   # default: 9 frames per launch x 3 lanes, 200 steps
python bench.py --pair 1 --no-cpu-baseline --no-batched --no-configs > "$OUT/bench_pair1.json" 2>> "$OUT/bench.err"
python bench.py --pregather --main-only --no-cpu-baseline > "$OUT/bench_pregather.json" 2>> "$OUT/bench.err"      # round 6 opt-in (cam_pregather)
python bench.py --train --steps 50 --warmup 5 > "$OUT/bench_train.json" 2>> "$OUT/bench.err"
python bench.py --train --deterministic --steps 50 --warmup 5 --no-live-pmc > "$OUT/bench_train_det.json" 2>> "$OUT/bench.err"   # order-free backward
cd /tmp
Q="--main-only --min-window-s 0.05 --warmup-s 0.05"   # the timed loop only, short windows: small traces
# per-kernel durations, one launch sequence at a time (with frames in flight the kernels of different lanes share the GPU)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$REPO/bench.py" --lanes 1 --pair 9 --steps 54 --warmup 9 $Q > "$OUT/prof.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_pair1" -- python3 "$REPO/bench.py" --lanes 1 --pair 1 --steps 50 --warmup 5 $Q > "$OUT/prof_pair1.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_f32" -- python3 "$REPO/bench.py" --matrix-path f32 --lanes 1 --pair 9 --steps 54 --warmup 9 $Q > "$OUT/prof_f32.log" 2>&1
# the default command (3 lanes): trace of the overlap
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_lanes" -- python3 "$REPO/bench.py" --pair 9 --steps 54 --warmup 9 $Q > "$OUT/prof_lanes.log" 2>&1
for B in 9 1; do
  for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_${C}_b$B" -- python3 "$REPO/bench.py" --batch $B --steps 5 --warmup 2 $Q --no-graph > "$OUT/pmc_${C}_b$B.log" 2>&1
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_train" -- python3 "$REPO/bench.py" --train --steps 20 --warmup 3 > "$OUT/prof_train.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_pregather" -- python3 "$REPO/bench.py" --pregather --lanes 1 --pair 9 --steps 54 --warmup 9 $Q > "$OUT/prof_pregather.log" 2>&1
cd "$REPO"
python tools/dropin_breakdown.py > "$OUT/dropin_breakdown.txt" 2>&1
[ -x tools/pair_exchange_probe ] && timeout 120 tools/pair_exchange_probe > "$OUT/pair_exchange_probe.txt" 2>&1
# keep the merge-back small (gpurun copies back at most 64 MiB): stats / trace / counter csv only
find "$OUT" -type f ! -name '*kernel_stats.csv' ! -name '*kernel_trace.csv' ! -name '*counter_collection.csv' \
     ! -name '*.json' ! -name '*.log' ! -name '*.err' ! -name '*.txt' -delete 2>/dev/null
find "$OUT" -name '*kernel_trace.csv' -size +12M -delete 2>/dev/null
# the counter csv of a PMC pass carries every dispatch of the process: keep the kernels of the path
for f in $(find "$OUT" -name '*counter_collection.csv'); do
  (head -1 "$f"; grep -E 'chain_|self_attn|box_decode|nchw_to_nhwc|radar_' "$f") > "$f.tmp" && mv "$f.tmp" "$f"
done
du -sh "$OUT"; du -s "$OUT"/* | sort -n | tail -12
[ -f "$OUT/pytest_gpu.log" ] && tail -3 "$OUT/pytest_gpu.log"; cat "$OUT/bench.json" | cut -c1-600
