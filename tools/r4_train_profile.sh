#!/bin/bash
# the training part of tools/profile_round.sh alone (after a change that only touches the training iteration)
TAG=${1:-r4t}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
python bench.py --train --steps 50 --warmup 5 > "$OUT/bench_train.json" 2> "$OUT/bench.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_train" -- python3 "$REPO/bench.py" --train --steps 20 --warmup 3 > "$OUT/prof_train.log" 2>&1
cd "$REPO"
find "$OUT" -type f ! -name '*kernel_stats.csv' ! -name '*.json' ! -name '*.log' ! -name '*.err' -delete 2>/dev/null
cat "$OUT/bench_train.json" | cut -c1-300
