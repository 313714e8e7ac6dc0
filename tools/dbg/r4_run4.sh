mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_gpu_train.py tests/test_gpu_loop.py tests/test_ddp_gloo.py -q -s -k "beside or injected or layerwise or two_rank or bench_launches or loop or ddp" > gpurun_out/t4.log 2>&1; echo "pytest rc $?" >> gpurun_out/t4.log
grep -n "bucket MB\|relL2\|undisturbed\|beside\|re-run\|single graph" gpurun_out/t4.log | head -20
tail -8 gpurun_out/t4.log
