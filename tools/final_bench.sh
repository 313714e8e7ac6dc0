#!/bin/bash
# Every committed number of a round from ONE gpurun call on the final state (run from the repo root on the GPU box):
#   tools/final_bench.sh r5   ->  gpurun_out/r5_bench_{fsq,gow,stock,big}.json, *_step_{seq,summary}.txt, *_kernel_stats.csv,
#                                 r5_force_comm_{fp32,bf16}.json; copy what is to be judged into profiles/
tag=${1:-rX}
set -x
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench_fsq.json 2> gpurun_out/${tag}_bench_fsq.err
bash tools/prof_step.sh ${tag}_bench_fsq > /dev/null
python bench.py --workload gow --no-live-pmc --no-sub > gpurun_out/${tag}_bench_gow.json 2> gpurun_out/${tag}_bench_gow.err
bash tools/prof_step.sh ${tag}_bench_gow --workload gow --no-tail > /dev/null
python bench.py --variant stock --no-live-pmc --no-sub > gpurun_out/${tag}_bench_stock.json 2> gpurun_out/${tag}_bench_stock.err
bash tools/prof_step.sh ${tag}_bench_stock --variant stock > /dev/null
python bench.py --workload big --steps 30 --warmup 5 --no-live-pmc --no-sub > gpurun_out/${tag}_bench_big.json 2> gpurun_out/${tag}_bench_big.err
bash tools/prof_step.sh ${tag}_bench_big --workload big --steps 12 --warmup 4 > /dev/null
for gc in fp32 bf16; do
  timeout 600 python bench.py --force-comm --grad-comm $gc --no-cpu-baseline --no-stress > gpurun_out/${tag}_force_comm_$gc.json 2> gpurun_out/${tag}_force_comm_$gc.err
done
mkdir -p gpurun_out/${tag}_pmc && cp gpurun_out/pmc_live/*.csv gpurun_out/${tag}_pmc/ 2>/dev/null
python - <<PY
import json
for n in ("fsq", "gow", "stock", "big"):
    try:
        j = json.load(open("gpurun_out/${tag}_bench_%s.json" % n))
        print(n, round(j["value"], 1), round(j["ms_per_step"], 4), (j.get("value_with_collate") or {}).get("value"), (j.get("roofline_stress") or {}).get("frac"),
              (j.get("roofline_stress_bwd") or {}).get("frac"), (j.get("roofline_gow_tail") or {}).get("step_ms"))
    except Exception as e:
        print(n, "ERR", e)
for gc in ("fp32", "bf16"):
    try:
        j = json.load(open("gpurun_out/${tag}_force_comm_%s.json" % gc))
        print("force-comm", gc, round(j["value"], 1), round(j["ms_per_step"], 4), j["comm_backend"], j["rccl_ranks"], j["allreduce_exposed_us"], j["ddp_one_graph"])
    except Exception as e:
        print("force-comm", gc, "ERR", e)
PY
