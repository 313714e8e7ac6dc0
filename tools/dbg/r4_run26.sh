mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/t26.log 2>&1; echo "pytest rc $?" >> gpurun_out/t26.log
tail -3 gpurun_out/t26.log
for w in fsq gow; do
python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub > gpurun_out/b26_$w.json 2> gpurun_out/b26_$w.err
done
python bench.py --variant stock --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub > gpurun_out/b26_stock.json 2> gpurun_out/b26_stock.err
python bench.py --workload big --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub --no-parity > gpurun_out/b26_big.json 2> gpurun_out/b26_big.err
python - <<PY
import json
for n in ("fsq","gow","stock","big"):
    j=json.load(open('gpurun_out/b26_%s.json'%n)); print(n, j["value"], j["ms_per_step"], (j.get("parity") or {}).get("worst_max_abs_logit_err"))
PY
