#!/bin/bash
# round 4: the whole GPU suite, the batch-vs-single diagnosis twice, the main timed loop on both matrix paths
out=gpurun_out/${1:-r4c}
mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $out/pytest_gpu.txt
tail -4 $out/pytest_gpu.txt
for i in 1 2; do timeout 300 python tools/r4_diag2.py 8 2>&1 | grep -E "compact|f16x2 (enc|again|third)|f32 K" ; done > $out/diag2.txt
grep -c "rows with cls diff > 1e-3: 0" $out/diag2.txt; grep -v "rows with cls diff > 1e-3: 0" $out/diag2.txt | grep -v compact | head
timeout 300 python tools/r4_diag.py res101 2>&1 | grep -E "identical|cls max" > $out/diag1.txt; cat $out/diag1.txt | head -12
for mp in f16x2 f32; do
  timeout 300 python bench.py --steps 200 --warmup 20 --main-only --no-cpu-baseline --matrix-path $mp > $out/bench_$mp.json 2> $out/bench_$mp.err
  python -c "import json,sys; d=json.loads(open('$out/bench_$mp.json').read().strip().splitlines()[-1]); print('$mp', d['value'], d['ms_per_step'])"
done
