mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/t9.log 2>&1; echo "pytest rc $?" >> gpurun_out/t9.log
tail -5 gpurun_out/t9.log
bash tools/prof_step.sh r4_bench_fsq
bash tools/prof_step.sh r4_bench_stock --variant stock
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_fsq.json 2> gpurun_out/r4_bench_fsq.err ) 2> gpurun_out/r4_bench_fsq.time
tail -2 gpurun_out/r4_bench_fsq.time
python - <<'PY'
import json
j=json.load(open('gpurun_out/r4_bench_fsq.json'))
print("fsq", round(j["value"]), j["ms_per_step"], "long", j["long_run"]["value"], "loop", j["value_with_collate"].get("value"), "stress", j["roofline_stress"]["frac"], j["roofline_stress_bwd"]["frac"])
for k,v in j["workloads"].items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("in_step"))
PY
