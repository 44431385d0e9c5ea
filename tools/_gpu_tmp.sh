python -m pytest tests -x -q -m gpu > gpurun_out/r6_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r6_pytest_gpu.log; tail -4 gpurun_out/r6_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
SKIP_TESTS=1 bash tools/profile_round.sh r6 > gpurun_out/r6_profile.log 2>&1; tail -2 gpurun_out/r6_profile.log | cut -c1-300
