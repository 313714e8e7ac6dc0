# Round 4: every committed number from ONE gpurun call on the final state (run from the repo root on the GPU box).
set -x
mkdir -p gpurun_out
python bench.py > gpurun_out/r4_bench_fsq.json 2> gpurun_out/r4_bench_fsq.err
bash tools/prof_step.sh r4_bench_fsq > /dev/null
python bench.py --workload gow --no-live-pmc --no-sub > gpurun_out/r4_bench_gow.json 2> gpurun_out/r4_bench_gow.err
bash tools/prof_step.sh r4_bench_gow --workload gow > /dev/null
python bench.py --variant stock --no-live-pmc --no-sub > gpurun_out/r4_bench_stock.json 2> gpurun_out/r4_bench_stock.err
bash tools/prof_step.sh r4_bench_stock --variant stock > /dev/null
python bench.py --workload big --steps 30 --warmup 5 --no-live-pmc --no-sub > gpurun_out/r4_bench_big.json 2> gpurun_out/r4_bench_big.err
bash tools/prof_step.sh r4_bench_big --workload big --steps 12 --warmup 4 > /dev/null
ls gpurun_out/pmc_live gpurun_out/sub_big 2>/dev/null | head -30
python - <<PY
import json
for n in ("fsq", "gow", "stock", "big"):
    j = json.load(open("gpurun_out/r4_bench_%s.json" % n))
    print(n, j["value"], j["ms_per_step"], j.get("value_with_collate"), (j.get("roofline_stress") or {}).get("frac"), (j.get("roofline_stress_bwd") or {}).get("frac"))
PY
