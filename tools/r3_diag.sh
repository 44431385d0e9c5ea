#!/bin/bash
# Round-3 diagnostics on an MI355X box:   gpurun --timeout 1500 -- 'bash tools/r3_diag.sh'
# -> gpurun_out/r3diag/: GPU test log, per-step s_memtime stamps of the 16-row chains at 8 frames per
# launch (STAMPS build), and the main timed loop at several frames-per-launch / window settings.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r3diag
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
  tail -3 "$OUT/pytest_gpu.log"
fi
ST=$REPO/build/hip_stamps/libtranscar_hip_stamps.so
if [ -f "$ST" ]; then
  for B in ${STAMP_BATCHES-8}; do
    TRANSCAR_ALLOW_STAMPS=1 TRANSCAR_HIP_LIB=$ST STAMPS_BATCH=$B timeout 300 python tools/chain_stamps.py decoder > "$OUT/stamps_dec_b$B.txt" 2>&1
    TRANSCAR_ALLOW_STAMPS=1 TRANSCAR_HIP_LIB=$ST STAMPS_BATCH=$B timeout 300 python tools/chain_stamps.py radar > "$OUT/stamps_rad_b$B.txt" 2>&1
  done
fi
Q="--main-only"
: > "$OUT/bench_sweep.jsonl"
for cfg in ${SWEEP-8:200:20 9:180:20 9:20:5 7:20:5 10:20:5}; do
  IFS=: read -r a b c <<< "$cfg"; set -- $a $b $c
  echo "# pair $1 steps $2 warmup $3" >> "$OUT/bench_sweep.jsonl"
  timeout 300 python bench.py --pair $1 --steps $2 --warmup $3 $Q >> "$OUT/bench_sweep.jsonl" 2>> "$OUT/bench.err"
done
for P in ${ROOF-9}; do
  timeout 300 python bench.py --pair $P --steps $((P * 20)) --no-cpu-baseline --no-batched --no-handoff > "$OUT/bench_roof_p$P.json" 2>> "$OUT/bench.err"
done
python - <<'PY'
import json, os
p = os.path.join('gpurun_out', 'r3diag', 'bench_sweep.jsonl')
for l in open(p):
    if l.startswith('#'):
        print(l.strip(), end='  ')
    elif l.startswith('{'):
        d = json.loads(l)
        print('%.0f frames/s  (fpl %s, window %.2f ms)' % (d['value'], d['config']['frames_per_launch'], d['timing']['window_ms_median']))
PY
