mkdir -p gpurun_out/r6f
python -m pytest tests/test_gpu_training.py tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r6f/pytest.log 2>&1; tail -8 gpurun_out/r6f/pytest.log
python bench.py --train --steps 50 --warmup 5 --no-live-pmc > gpurun_out/r6f/bench_train.json 2> gpurun_out/r6f/bench_train.err
python -c "
import json
d=json.loads(open('gpurun_out/r6f/bench_train.json').read().strip().splitlines()[-1])
print('train', d['value'], d['ms_per_step'], d['exposed_collective_ms'])
"
python bench.py --train --gpus 2 --share-gpu --backend gloo --steps 20 --warmup 3 --min-window-s 0.2 --warmup-s 0.1 --no-live-pmc > gpurun_out/r6f/bench_train_2ranks.json 2> gpurun_out/r6f/bench_train_2ranks.err
python -c "
import json
d=json.loads(open('gpurun_out/r6f/bench_train_2ranks.json').read().strip().splitlines()[-1])
print('train 2 ranks (shared GPU, gloo)', d['value'], d['ms_per_step'], d['exposed_collective_ms'])
"
