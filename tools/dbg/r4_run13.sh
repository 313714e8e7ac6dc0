mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_c5.py tests/test_gpu_kernels.py tests/test_gpu_scale.py tests/test_gpu_layer.py tests/test_gpu_train_parity.py tests/test_gpu_bench_parity.py tests/test_gpu_loop.py -x -q > gpurun_out/t13.log 2>&1; echo "pytest rc $?" >> gpurun_out/t13.log
tail -6 gpurun_out/t13.log
for v in 0 1; do
MOBGT_ATTN_TWO_PASS=$v python bench.py --workload big --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub --no-parity > gpurun_out/b_big1p_$v.json 2> gpurun_out/b_big1p_$v.err
python - <<PY
import json
j=json.load(open('gpurun_out/b_big1p_$v.json'))
print("big two_pass=$v", j["value"], j["ms_per_step"], j["final_loss"])
PY
done
MOBGT_ATTN_TWO_PASS=0 python bench.py --workload gow --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub > gpurun_out/b_gow1p.json 2> gpurun_out/b_gow1p.err
python - <<PY
import json
j=json.load(open('gpurun_out/b_gow1p.json'))
print("gow one-pass", j["value"], j["ms_per_step"], j["parity"]["worst_max_abs_logit_err"])
PY
