mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_distgcn.py -x -q -s > gpurun_out/t14.log 2>&1; echo "pytest rc $?" >> gpurun_out/t14.log
tail -30 gpurun_out/t14.log
