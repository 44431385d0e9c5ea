#!/bin/bash
# kernel stats of the training iteration: gpurun --timeout 900 -- 'bash tools/r3_trainprof.sh'
R=$(pwd); OUT=$R/gpurun_out/r3train; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --train --steps 20 --warmup 3 > $OUT/prof.log 2>&1
cd $R
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n_it = None
for r in rows:
    if 'adamw_kernel' in r['Name']:
        n_it = int(r['Calls'])
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
print('iterations %s, launches/iter %.1f, kernel time/iter %.1f us' % (n_it, calls / n_it, tot / n_it / 1e3))
for r in rows[:28]:
    print('%-70s calls/it %5.1f  us/it %7.1f  avg %6.1f us' % (r['Name'][:70], int(r['Calls']) / n_it, float(r['TotalDurationNs']) / n_it / 1e3, float(r['AverageNs']) / 1e3))
PY
find $OUT -type f ! -name '*kernel_stats.csv' ! -name '*.log' -delete 2>/dev/null
tail -1 $OUT/prof.log | cut -c1-300
