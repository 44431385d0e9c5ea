mkdir -p gpurun_out/r6l
Q="--main-only --no-cpu-baseline"
for cfg in "--lanes 3" "--tile-rows 16 --lanes 3" "--tile-rows 16 --lanes 4" "--tile-rows 16 --lanes 6" "--lanes 4" "--lanes 6" "--tile-rows 16 --lanes 6 --pair 5" "--tile-rows 16 --lanes 8 --pair 4"; do
  python bench.py --steps 216 --warmup 18 $Q $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('216 [$cfg]', round(d['value'],1))"
  GPU_MAX_HW_QUEUES=8 python bench.py --steps 216 --warmup 18 $Q $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('216 [$cfg] 8 hw queues', round(d['value'],1))"
done > gpurun_out/r6l/lanes16.txt 2>&1
cat gpurun_out/r6l/lanes16.txt
