mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_layer.py tests/test_gpu_train_parity.py "tests/test_gpu_bench_parity.py::test_stock_variant_of_the_bench_vs_oracle_and_through_trainstep" -x -q -s -k "preln or stock or g4 or chain" > gpurun_out/t5.log 2>&1; echo "pytest rc $?" >> gpurun_out/t5.log
tail -12 gpurun_out/t5.log
python bench.py --variant stock --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub > gpurun_out/b_stock.json 2> gpurun_out/b_stock.err; tail -3 gpurun_out/b_stock.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/b_stock.json'))
print("stock", j["value"], j["ms_per_step"], j["parity"]["worst_max_abs_logit_err"] if j.get("parity") else None)
PY
