#!/bin/bash
# end-to-end frame rate with variants of the staged attention kernel (tools/sx_variants.sh)
for v in "$@"; do
  TRANSCAR_HIP_LIB=$PWD/build/sx/lib_$v.so timeout 300 python bench.py --steps 200 --warmup 20 --main-only --no-cpu-baseline > /tmp/b_$v.json 2>/tmp/b_$v.err
  python -c "import json; d=json.loads(open('/tmp/b_$v.json').read().strip().splitlines()[-1]); print('$v', round(d['value'],1), d['ms_per_step'])" || tail -3 /tmp/b_$v.err
done
