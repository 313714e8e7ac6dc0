mkdir -p gpurun_out
ROUNDS=3 tools/r4_ab.sh mad24 late emask all3 > gpurun_out/ab2.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_scale.py tests/test_gpu_train_parity.py "tests/test_gpu_train.py::test_parameter_reached_twice_per_pass_gets_no_gradient_sink" "tests/test_gpu_train.py::test_bench_launches_two_ranks_and_reports_them" -q -s -k "not fq_layer_with_dropout" > gpurun_out/t2.log 2>&1; echo "pytest rc $?" >> gpurun_out/t2.log
cat gpurun_out/ab2.log | tail -8; tail -5 gpurun_out/t2.log
