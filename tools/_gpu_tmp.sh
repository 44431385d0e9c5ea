mkdir -p gpurun_out/r6e
Q="--main-only --no-cpu-baseline"
for i in 1 2 3; do
for fl in "" "--no-pregather"; do
  python bench.py --gpus 1 --steps 20 --warmup 5 $Q $fl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver [$fl]', round(d['value'],1))"
  python bench.py --steps 216 --warmup 18 $Q $fl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('216   [$fl]', round(d['value'],1))"
done; done > gpurun_out/r6e/ab.txt 2>&1
cat gpurun_out/r6e/ab.txt
