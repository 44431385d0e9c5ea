#!/bin/bash
# stagger experiment: gpurun --timeout 900 -- 'bash tools/r3_stagger.sh "0 20000 40000" 8'
set -u
OUT=gpurun_out/r3stag; mkdir -p $OUT
P=${2:-8}
for D in $1; do
  TRANSCAR_STAGGER=$D timeout 300 python bench.py --pair $P --steps $((P * 25)) --no-cpu-baseline --no-batched --no-handoff > $OUT/s_${P}_$D.json 2>> $OUT/err.log
  python - $OUT/s_${P}_$D.json $D <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d['roofline']; o = r['others']
    print('stagger %6s: %.0f frames/s  chain %.1f us (%.3f)  radar %.1f us  attn %.1f us  latency %.3f ms' % (
        sys.argv[2], d['value'], r['ms'] * 1e3, r['frac'], o['chain_kernel(radar fusion)']['ms'] * 1e3,
        o['self_attn_kernel']['ms'] * 1e3, d['latency_ms_per_frame']))
except Exception as e:
    print('stagger', sys.argv[2], 'failed', e)
PY
done
tail -3 $OUT/err.log
