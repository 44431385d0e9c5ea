#!/bin/bash
# round 4, first GPU pass of the two-plane f16 matrix path: parity tests that exercise it, then the main timed
# loop on both matrix paths (same box, back to back)
mkdir -p gpurun_out/r4b
timeout 1500 python -m pytest tests/test_gpu_teacher_forced.py -x -q -m gpu -s 2>&1 | tail -60 > gpurun_out/r4b/teacher.txt
tail -5 gpurun_out/r4b/teacher.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "frame_in_a_batch or full_size_launch or timed_geometry or row_order" 2>&1 | tail -15 > gpurun_out/r4b/parity_subset.txt
tail -5 gpurun_out/r4b/parity_subset.txt
for mp in f16x2 f32; do
  timeout 300 python bench.py --steps 200 --warmup 20 --main-only --no-cpu-baseline --matrix-path $mp > gpurun_out/r4b/bench_$mp.json 2> gpurun_out/r4b/bench_$mp.err
  cat gpurun_out/r4b/bench_$mp.json | cut -c1-400
done
