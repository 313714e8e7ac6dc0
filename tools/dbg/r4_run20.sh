mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/t20.log 2>&1; echo "pytest rc $?" >> gpurun_out/t20.log
tail -4 gpurun_out/t20.log
for w in fsq gow; do
python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub > gpurun_out/b20_$w.json 2> gpurun_out/b20_$w.err
done
python - <<PY
import json
for n in ("fsq","gow"):
    j=json.load(open('gpurun_out/b20_%s.json'%n)); print(n, j["value"], j["ms_per_step"], j["parity"]["worst_max_abs_logit_err"])
PY
