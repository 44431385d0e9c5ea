#!/bin/bash
out=gpurun_out/${1:-r4i}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "lookahead or prefetch or many_radar or dropout" 2>&1 | tail -15
for d in 1 9; do
  timeout 300 python bench.py --train --steps 100 --warmup 20 --prefetch-depth $d --no-cpu-baseline > $out/train_d$d.json 2> $out/train_d$d.err
  python - <<PY
import json
d = json.loads(open('$out/train_d$d.json').read().strip().splitlines()[-1])
print('depth $d', 'ms/iter', d['ms_per_step'], 'parts', {k[:22]: (round(v['ms'], 3), round(v['frac'], 3)) for k, v in d['roofline']['parts'].items()})
PY
  tail -2 $out/train_d$d.err
done
