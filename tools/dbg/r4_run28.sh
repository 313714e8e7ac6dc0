for i in 1 2; do
python bench.py --steps 100 --warmup 10 --no-stress --no-live-pmc --no-sub > gpurun_out/b28_$i.json 2> gpurun_out/b28_$i.err
python - <<PY
import json
j=json.load(open('gpurun_out/b28_$i.json')); v=j["value_with_collate"]
print("run $i", j["value"], j["ms_per_step"], v["value"], v["ms_per_step"], v["ms_per_step_same_graphs_no_input"], v["ms_per_step_by_epoch_host_side"])
PY
done
