mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/t3.log 2>&1; echo "pytest rc $?" >> gpurun_out/t3.log
tail -15 gpurun_out/t3.log
