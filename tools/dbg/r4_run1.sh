mkdir -p gpurun_out
tools/r4_ab.sh tiled > gpurun_out/ab1.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_scale.py tests/test_gpu_train_parity.py -x -q -s -k "not fq_layer_with_dropout" > gpurun_out/t1.log 2>&1; echo "pytest rc $?" >> gpurun_out/t1.log
( time python bench.py --steps 20 --warmup 5 > gpurun_out/b_default.json 2> gpurun_out/b_default.err ) 2> gpurun_out/b_default.time
tail -3 gpurun_out/ab1.log; tail -3 gpurun_out/t1.log; cat gpurun_out/b_default.time
