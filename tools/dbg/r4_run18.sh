mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/t18.log 2>&1; echo "pytest rc $?" >> gpurun_out/t18.log
tail -8 gpurun_out/t18.log
python bench.py --workload gow --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub > gpurun_out/b18_gow.json 2> gpurun_out/b18_gow.err
python - <<PY
import json
j=json.load(open('gpurun_out/b18_gow.json')); print("gow", j["value"], j["ms_per_step"], j["parity"]["worst_max_abs_logit_err"])
PY
