# three default bench runs back to back: the spread of the headline and its host_stalls
mkdir -p gpurun_out
for i in 1 2 3; do
python bench.py > gpurun_out/r4_default_$i.json 2> gpurun_out/r4_default_$i.err
python - <<PY
import json
j = json.load(open("gpurun_out/r4_default_$i.json"))
print($i, round(j["value"], 1), round(j["ms_per_step"], 4), j["ms_per_step_chunks"], j["host_stalls"], (j.get("value_with_collate") or {}).get("ms_per_step"))
PY
done
