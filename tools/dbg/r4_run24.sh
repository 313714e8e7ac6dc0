mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/t24.log 2>&1; echo "pytest rc $?" >> gpurun_out/t24.log
tail -3 gpurun_out/t24.log
