mkdir -p gpurun_out
for v in 0 1; do
MOBGT_CHAIN_BIG=$v python bench.py --workload big --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub --no-parity > gpurun_out/b_big_$v.json 2> gpurun_out/b_big_$v.err
python - <<PY
import json
j=json.load(open('gpurun_out/b_big_$v.json'))
print("big chain_big=$v", j["value"], j["ms_per_step"], j["final_loss"])
PY
done
