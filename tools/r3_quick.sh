#!/bin/bash
# quick check on the GPU box: parity tests (optionally a -k subset) + the bench at one setting
#   gpurun --timeout 900 -- 'bash tools/r3_quick.sh "<pytest -k expr or empty>" "<bench args>"'
set -u
OUT=gpurun_out/r3quick; mkdir -p $OUT
if [ "${1:-}" != "skip" ]; then
  if [ -n "${1:-}" ]; then python -m pytest tests -x -q -m gpu -k "$1" > $OUT/pytest.log 2>&1; else python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; fi
  echo "pytest rc=$?"; tail -15 $OUT/pytest.log
fi
shift
for args in "$@"; do
  timeout 300 python bench.py $args --no-cpu-baseline --no-batched --no-handoff > $OUT/bench.json 2>> $OUT/err.log
  python - $OUT/bench.json "$args" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get('roofline'); 
    msg = '%s: %.0f frames/s' % (sys.argv[2], d['value'])
    if r:
        o = r['others']
        msg += '  chain %.1f us (%.3f)  radar %.1f us  attn %.1f us' % (r['ms'] * 1e3, r['frac'], o['chain_kernel(radar fusion)']['ms'] * 1e3, o['self_attn_kernel']['ms'] * 1e3)
    if 'latency_ms_per_frame' in d:
        msg += '  latency %.3f ms' % d['latency_ms_per_frame']
    print(msg)
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
done
tail -3 $OUT/err.log 2>/dev/null
