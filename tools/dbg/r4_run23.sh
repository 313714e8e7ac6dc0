mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/t23.log 2>&1; echo "pytest rc $?" >> gpurun_out/t23.log
tail -5 gpurun_out/t23.log
for w in fsq gow; do
python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub > gpurun_out/b23_$w.json 2> gpurun_out/b23_$w.err
done
MOBGT_NO_TOKEN_FWD_CHAIN=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub > gpurun_out/b23_fsq_split.json 2> gpurun_out/b23_fsq_split.err
python - <<PY
import json
for n in ("fsq","gow","fsq_split"):
    j=json.load(open('gpurun_out/b23_%s.json'%n)); print(n, j["value"], j["ms_per_step"], j["parity"]["worst_max_abs_logit_err"])
PY
