mkdir -p gpurun_out/r6o
Q="--main-only --no-cpu-baseline"
for i in 1 2 3; do
for cfg in "" "--pregather" "--weight-prefetch" "--pregather --weight-prefetch" "QNT"; do
  lib=""; fl="$cfg"; if [ "$cfg" = "QNT" ]; then lib="build/hip_qnt/libtranscar_hip_qnt.so"; fl=""; fi
  TRANSCAR_HIP_LIB=$lib python bench.py --gpus 1 --steps 20 --warmup 5 $Q $fl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver [$cfg]', round(d['value'],1))"
  TRANSCAR_HIP_LIB=$lib python bench.py --steps 216 --warmup 18 $Q $fl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('216   [$cfg]', round(d['value'],1))"
  TRANSCAR_HIP_LIB=$lib python bench.py --lanes 1 --steps 54 --warmup 9 $Q $fl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1lane [$cfg]', round(d['value'],1))"
done; done > gpurun_out/r6o/ab_optins.txt 2>&1
sort gpurun_out/r6o/ab_optins.txt
