#!/bin/bash
# final checks of a round on the GPU box: gpurun --timeout 2400 -- 'bash tools/r3_final.sh'
OUT=gpurun_out/r3final; mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
python bench.py --shapes vovnet --no-cpu-baseline --no-batched --no-handoff > $OUT/bench_vovnet.json 2> $OUT/err.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3final/bench_vovnet.json').read().strip().splitlines()[-1])
print('vovnet shapes: %.0f frames/s, fpl %s, latency %.3f ms, chain frac %.3f' % (d['value'], d['config']['frames_per_launch'], d['latency_ms_per_frame'], d['roofline']['frac']))
PY
# the two-rank code path with real kernels on the one GPU (gloo; RCCL refuses two ranks on one device)
python bench.py --gpus 2 --share-gpu --backend gloo --steps 20 --warmup 5 --main-only > $OUT/bench_2ranks_shared.json 2>> $OUT/err.log
python bench.py --gpus 2 --share-gpu --backend gloo --train --steps 20 --warmup 3 --no-roofline > $OUT/bench_2ranks_train.json 2>> $OUT/err.log
python - <<'PY'
import json
for f in ('bench_2ranks_shared', 'bench_2ranks_train'):
    try:
        d = json.loads(open('gpurun_out/r3final/%s.json' % f).read().strip().splitlines()[-1])
        print(f, 'n_gpus', d['n_gpus'], 'ranks', d['rccl_ranks'], 'value %.0f' % d['value'], 'per_rank', [round(x) for x in d['per_rank']['frames_per_s']], d['cpu_affinity'])
    except Exception as e:
        print(f, 'failed', e)
PY
tail -3 $OUT/err.log
