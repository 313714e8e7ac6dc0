mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_train.py tests/test_gpu_bench_parity.py tests/test_gpu_train_parity.py tests/test_gpu_loop.py tests/test_gpu_model.py tests/test_gpu_distgcn.py -x -q > gpurun_out/t21.log 2>&1; echo "pytest rc $?" >> gpurun_out/t21.log
tail -5 gpurun_out/t21.log
bash tools/prof_step.sh r4g_fsq > gpurun_out/prof21.log 2>&1
sed -n 1,12p gpurun_out/r4g_fsq_step_seq.txt | cut -c1-100
head -1 gpurun_out/r4g_fsq_step_summary.txt
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub > gpurun_out/b21_fsq.json 2> gpurun_out/b21_fsq.err
python - <<PY
import json
j=json.load(open('gpurun_out/b21_fsq.json')); print("fsq", j["value"], j["ms_per_step"], j["parity"]["worst_max_abs_logit_err"])
PY
