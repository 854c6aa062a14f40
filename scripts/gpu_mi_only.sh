mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_mi.py -m gpu -q -x > gpurun_out/pytest_mi.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_mi.log
tail -25 gpurun_out/pytest_mi.log
exit $rc
