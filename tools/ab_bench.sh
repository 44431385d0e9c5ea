#!/bin/bash
# Same-box A/B of two builds of the library: alternating runs of the default bench line, the driver's command and one
# launch sequence at a time, then the output hashes of both (tools/ab_hash.py).   bash tools/ab_bench.sh OLD.so [out.txt]
#   make -C transcar_amd/csrc -j8 BUILD=../../build/ab_old LIB=../../build/ab_old/libtranscar_old.so EXTRA=-D...
OLD=${1:-build/ab_old/libtranscar_old.so}
OUT=${2:-gpurun_out/ab.txt}
mkdir -p "$(dirname "$OUT")"
V='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"])'
{
echo "== hashes new"; python tools/ab_hash.py 2>&1 | grep "^rows"
echo "== hashes old"; TRANSCAR_HIP_LIB=$OLD python tools/ab_hash.py 2>&1 | grep "^rows"
for i in 1 2 3; do
  echo "new default: $(python bench.py --main-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$V")"
  echo "old default: $(TRANSCAR_HIP_LIB=$OLD python bench.py --main-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$V")"
  echo "new driver: $(python bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$V")"
  echo "old driver: $(TRANSCAR_HIP_LIB=$OLD python bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$V")"
  echo "new 1lane: $(python bench.py --lanes 1 --main-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$V")"
  echo "old 1lane: $(TRANSCAR_HIP_LIB=$OLD python bench.py --lanes 1 --main-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$V")"
done
} > "$OUT" 2>&1
