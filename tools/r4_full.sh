#!/bin/bash
# everything: the GPU suite, the driver's command (timed), the default bench, the training bench
out=gpurun_out/${1:-r4n}; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $out/pytest_gpu.txt; tail -3 $out/pytest_gpu.txt
s=$(date +%s); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; e=$(date +%s)
echo "driver command wall: $((e - s)) s"
python - <<PY
import json
d = json.loads(open('$out/bench_driver.json').read().strip().splitlines()[-1])
r = d['roofline']
print('value', round(d['value'], 1), '| dom', r['kernel'], 'us', round(r['ms'] * 1e3, 1), 'frac', round(r['frac'], 3), 'f32frac', round(r.get('frac_of_f32_mfma_peak', 0), 3), 'busy', r.get('mfma_busy'), 'traffic', r.get('traffic'))
print('f32_path', d.get('f32_path', {}).get('value'), '| vovnet', d.get('vovnet', {}).get('value'), '| train ms', d.get('train', {}).get('ms_per_iteration'), '| latency', d.get('latency_ms_per_frame'))
print('dropin', {k: (v.get('ms_per_frame') if isinstance(v, dict) else v) for k, v in d.get('dropin_forward', {}).items()})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['ms_per_frame'])
PY
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python -c "import json; d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); print('default (200 steps)', d['value'], d['roofline']['ms'], d['roofline']['path_frac'])"
timeout 300 python bench.py --train > $out/bench_train.json 2> $out/bench_train.err
python -c "import json; d=json.loads(open('$out/bench_train.json').read().strip().splitlines()[-1]); print('train', d['ms_per_step'], d['config'].get('decoder_lookahead_frames'))"
