#!/bin/bash
# which steps the 16-row chains' time goes to: the STAMPS build's TRANSCAR_CHAIN_DBG switches (wrong results, valid timings)
#   1 no linear epilogues, 2 no barriers, 4 no LayerNorm steps, 8 no camera sampling, 32 no linear item loops
#   gpurun --timeout 900 -- 'bash tools/r3_ablate.sh'
export TRANSCAR_ALLOW_STAMPS=1 TRANSCAR_HIP_LIB=build/hip_stamps/libtranscar_hip_stamps.so
OUT=gpurun_out/r3ablate; mkdir -p $OUT
for dbg in ${ABLATE-0 4 8 1 32 12 36 37 45}; do
  TRANSCAR_CHAIN_DBG=$dbg timeout 200 python bench.py --steps 90 --warmup 10 --no-cpu-baseline --no-batched --no-handoff > $OUT/b.json 2>> $OUT/err.log
  python - $OUT/b.json $dbg <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d['roofline']; o = r['others']
    print('dbg %3s: %5.0f frames/s  decoder chain %.1f us  radar chain %.1f us  attn %.1f us' % (sys.argv[2], d['value'], r['ms'] * 1e3, o['chain_kernel(radar fusion)']['ms'] * 1e3, o['self_attn_kernel']['ms'] * 1e3))
except Exception as e:
    print('dbg', sys.argv[2], 'failed', e)
PY
done
