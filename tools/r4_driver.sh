#!/bin/bash
# the driver's command (and the default), timed; the new regression test
out=gpurun_out/${1:-r4h}
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f16x2_chains" 2>&1 | tail -3
s=$(date +%s.%N)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
e=$(date +%s.%N)
echo "driver command wall: $(echo "$e - $s" | bc) s"; tail -3 $out/bench_driver.err
python - <<PY
import json
d = json.loads(open('$out/bench_driver.json').read().strip().splitlines()[-1])
print('value', d['value'], 'dtype', d['dtype'][:40])
r = d['roofline']
print('roofline', r['kernel'], 'ms', r['ms'], 'frac', r['frac'], 'peak', r['peak'], 'f32frac', r.get('frac_of_f32_mfma_peak'), 'traffic', r.get('traffic'), 'busy', r.get('mfma_busy'), 'wstream', r.get('weight_stream_gbs_per_cu'))
print('path_frac', r['path_frac'])
for k in ('f32_path', 'vovnet', 'train'):
    v = d.get(k); print(k, json.dumps(v)[:600] if v else None)
print('single', d.get('latency_ms_per_frame'), 'dropin', json.dumps(d.get('dropin_forward'))[:300])
print('cpu', json.dumps({k: d['cpu_baseline'][k] for k in ('value', 'cores', 'ms_per_frame', 'sample')}))
print('others', json.dumps(r['others'])[:900])
PY
