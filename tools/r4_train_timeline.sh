#!/bin/bash
# timeline of a few training iterations (kernel trace): where does the wall time of an iteration go?
#   gpurun --timeout 900 -- 'bash tools/r4_train_timeline.sh 9'
R=$(pwd); D=${1:-9}; OUT=$R/gpurun_out/r4train; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_d$D -- python3 $R/bench.py --train --steps 30 --warmup 10 --prefetch-depth $D --no-roofline --no-cpu-baseline --min-window-s 0.05 --warmup-s 0.05 > $OUT/prof_d$D.log 2>&1
cd $R
f=$(find $OUT/prof_d$D -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ends = [r for r in rows if 'adamw_kernel' in r['Kernel_Name']]
print('iterations traced:', len(ends))
# the last 12 iterations: wall per iteration (adamw end to adamw end), GPU busy (union of kernel intervals), per queue
sel = ends[-13:]
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
for a, b in zip(sel[:-1], sel[1:]):
    ks = [r for r in rows if r['s'] >= a['e'] and r['e'] <= b['e']]
    iv = sorted((r['s'], r['e']) for r in ks)
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None: busy += cur_e - cur_s
    perq = collections.Counter()
    for r in ks: perq[r[qkey]] += r['e'] - r['s']
    print('iteration %.0f us wall, %.0f us GPU busy (union), %d kernels; kernel time per queue %s' % (
        (b['e'] - a['e']) / 1e3, busy / 1e3, len(ks), {k: round(v / 1e3) for k, v in perq.items()}))
# one iteration in detail
a, b = sel[-3], sel[-2]
print('--- one iteration (us from the previous adamw end): start dur queue kernel')
for r in rows:
    if r['s'] >= a['e'] and r['e'] <= b['e'] + 1:
        print('%8.1f %7.1f  q%s  %s' % ((r['s'] - a['e']) / 1e3, (r['e'] - r['s']) / 1e3, r[qkey], r['Kernel_Name'][:90]))
PY
find $OUT/prof_d$D -type f -delete 2>/dev/null
tail -1 $OUT/prof_d$D.log | cut -c1-200
