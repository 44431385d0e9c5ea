#!/bin/bash
out=gpurun_out/${1:-r4l}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_teacher_forced.py -x -q -m gpu -k "sdpa or attention_core or decoder_layers_teacher or self_attn" 2>&1 | tail -8
python - <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
import bench; bench._imports()
from transcar_amd import ops
dev = torch.device('cuda:0')
for B in (1, 9):
    Q, C = 900, 256
    q = torch.randn(B, Q, C, device=dev) * 0.3; k = torch.randn(B, Q, C, device=dev); vt = torch.zeros(B, C, 912, device=dev); vt[:, :, :Q] = torch.randn(B, C, Q, device=dev)
    for mp in ('f32', 'f16x2'):
        for _ in range(5): o = ops.sdpa(q, k, vt, matrix_path=mp)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): o = ops.sdpa(q, k, vt, matrix_path=mp)
        e1.record(); torch.cuda.synchronize()
        print('B', B, mp, 'us per call (incl. cat + planes pass)', e0.elapsed_time(e1) / 50 * 1e3)
    a, b = ops.sdpa(q, k, vt, matrix_path='f32'), ops.sdpa(q, k, vt, matrix_path='f16x2')
    print('max diff f32 vs f16x2', float((a - b).abs().max()))
PY
for mp in f16x2 f32; do
  timeout 300 python bench.py --steps 200 --warmup 20 --main-only --no-cpu-baseline --matrix-path $mp > $out/bench_$mp.json 2> $out/bench_$mp.err
  python -c "import json,sys; d=json.loads(open('$out/bench_$mp.json').read().strip().splitlines()[-1]); print('$mp', d['value'], d['ms_per_step'])"
done
