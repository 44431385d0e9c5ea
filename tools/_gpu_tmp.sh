mkdir -p gpurun_out/r6h
python -m pytest tests -x -q -m gpu > gpurun_out/r6h/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r6h/pytest_gpu.log; tail -6 gpurun_out/r6h/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6h/smoke.log 2>&1; tail -3 gpurun_out/r6h/smoke.log
