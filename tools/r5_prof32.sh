set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r5i; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
Q="--main-only --min-window-s 0.05 --warmup-s 0.05"
for tr in 16 32; do
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$tr" -- python3 "$REPO/bench.py" --lanes 1 --pair 9 --steps 54 --warmup 9 --tile-rows $tr $Q > "$OUT/prof_$tr.log" 2>&1
f=$(find $OUT/prof_$tr -name '*kernel_stats.csv' | head -1); echo "== tile rows $tr"; head -12 "$f" | cut -c1-200
done
cd $REPO; find "$OUT" -type f ! -name '*kernel_stats.csv' ! -name '*.log' -delete 2>/dev/null
