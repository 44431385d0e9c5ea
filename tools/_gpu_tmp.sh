mkdir -p gpurun_out/r6q
Q="--main-only --no-cpu-baseline"
for i in 1 2 3; do
for lib in "" "build/hip_warm/libtranscar_hip_warm.so"; do
  TRANSCAR_HIP_LIB=$lib python bench.py --gpus 1 --steps 20 --warmup 5 $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver [$lib]', round(d['value'],1))"
  TRANSCAR_HIP_LIB=$lib python bench.py --steps 216 --warmup 18 $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('216   [$lib]', round(d['value'],1))"
  TRANSCAR_HIP_LIB=$lib python bench.py --lanes 1 --steps 54 --warmup 9 $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1lane [$lib]', round(d['value'],1))"
done; done > gpurun_out/r6q/ab_warm.txt 2>&1
sort gpurun_out/r6q/ab_warm.txt
