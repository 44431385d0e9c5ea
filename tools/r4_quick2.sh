#!/bin/bash
# diag (batch vs single, both paths) + the timed loop on both matrix paths; no test suite
out=gpurun_out/${1:-r4d}
mkdir -p $out
for i in 1 2 3; do timeout 300 python tools/r4_diag2.py 8 2>&1 | grep -E "compact|f16x2 (enc|again|third)|f32 K" ; done > $out/diag2.txt
echo "exact lines: $(grep -c 'rows with cls diff > 1e-3: 0' $out/diag2.txt) of $(grep -vc compact $out/diag2.txt)"
timeout 300 python tools/r4_diag.py res101 2>&1 | grep -E "identical|cls max" > $out/diag1.txt; grep -c "0.0, 0.0, 0.0" $out/diag1.txt; grep identical $out/diag1.txt
for mp in f16x2 f32 f16x2; do
  timeout 300 python bench.py --steps 200 --warmup 20 --main-only --no-cpu-baseline --matrix-path $mp > $out/bench_$mp.json 2> $out/bench_$mp.err
  python -c "import json,sys; d=json.loads(open('$out/bench_$mp.json').read().strip().splitlines()[-1]); print('$mp', d['value'], d['ms_per_step'])"
done
