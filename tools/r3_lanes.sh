#!/bin/bash
# lanes x frames-per-launch x hardware queues: gpurun --timeout 900 -- 'bash tools/r3_lanes.sh'
OUT=gpurun_out/r3lanes; mkdir -p $OUT; : > $OUT/res.txt
for cfg in "4 9 3 200" "8 9 3 200" "8 9 4 200" "8 9 5 200" "8 5 5 200" "8 4 6 200" "8 3 8 200" "8 9 4 20" "8 5 4 20" "8 7 3 20"; do
  set -- $cfg
  r=$(GPU_MAX_HW_QUEUES=$1 timeout 300 python bench.py --pair $2 --lanes $3 --steps $4 --warmup 5 --main-only 2>>$OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])")
  echo "queues $1 pair $2 lanes $3 steps $4: $r frames/s" | tee -a $OUT/res.txt
done
