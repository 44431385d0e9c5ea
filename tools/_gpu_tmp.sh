mkdir -p gpurun_out/r6k
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "plugin or pregather or pipeline or frames_in_flight or range_guard" > gpurun_out/r6k/pytest.log 2>&1; tail -4 gpurun_out/r6k/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6k/bench_driver.json 2> gpurun_out/r6k/bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r6k/bench_driver.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d.get('dropin_forward'))[200:1500]); print(d.get('single_lane'), d.get('latency_ms_per_frame'))
"
python tools/dropin_breakdown.py > gpurun_out/r6k/dropin_breakdown.txt 2>&1; sed -n 2,8p gpurun_out/r6k/dropin_breakdown.txt; tail -1 gpurun_out/r6k/dropin_breakdown.txt
