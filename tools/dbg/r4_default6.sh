# six default bench runs back to back (main leg + loop leg only): the spread of the headline and its host_stalls
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
python bench.py --no-sub --no-live-pmc --no-stress --no-cpu-baseline > gpurun_out/r4_soak_$i.json 2> gpurun_out/r4_soak_$i.err
python - <<PY
import json
j = json.load(open("gpurun_out/r4_soak_$i.json"))
print($i, round(j["value"], 1), round(j["ms_per_step"], 4), j["ms_per_step_chunks"], j["host_stalls"]["max_step_gap_ms"], j["host_stalls"]["gaps_over_threshold"], (j.get("value_with_collate") or {}).get("ms_per_step"))
PY
done
