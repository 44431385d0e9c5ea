OUT=gpurun_out/r3lanes; mkdir -p $OUT
for cfg in "4 9 3 180" "8 9 3 180" "8 9 4 180" "8 9 5 180" "4 9 4 180" "4 9 3 20" "8 9 3 20" "8 9 4 20" "8 7 3 20" "8 5 4 20"; do
  set -- $cfg
  r=$(GPU_MAX_HW_QUEUES=$1 timeout 200 python bench.py --pair $2 --lanes $3 --steps $4 --warmup 5 --main-only 2>>$OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])")
  echo "queues $1 pair $2 lanes $3 steps $4: $r frames/s"
done
