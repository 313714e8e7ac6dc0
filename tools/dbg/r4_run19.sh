mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_distgcn.py tests/test_gpu_train.py tests/test_gpu_bench_parity.py tests/test_gpu_train_parity.py tests/test_gpu_loop.py tests/test_gpu_model.py -x -q > gpurun_out/t19.log 2>&1; echo "pytest rc $?" >> gpurun_out/t19.log
tail -5 gpurun_out/t19.log
bash tools/prof_step.sh r4f_fsq > gpurun_out/prof19.log 2>&1
grep -n "mask_\|sgemm" gpurun_out/r4f_fsq_step_seq.txt | cut -c1-100
python - <<PY
import json
j=json.load(open('gpurun_out/r4f_fsq_bench.json')); print("fsq under profiler", j["value"], j["ms_per_step"])
PY
