mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_loop.py -x -q > gpurun_out/t7.log 2>&1; echo "pytest rc $?" >> gpurun_out/t7.log
tail -5 gpurun_out/t7.log
for v in 0 1; do
MOBGT_LOOP_D2D=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-live-pmc --no-sub --no-parity > gpurun_out/b_loop_$v.json 2> gpurun_out/b_loop_$v.err
python - <<PY
import json
j=json.load(open('gpurun_out/b_loop_$v.json'))
w=j["value_with_collate"]
print("d2d=$v value", round(j["value"]), "ms", round(j["ms_per_step"],4), "| loop", {k: (round(w[k],4) if isinstance(w[k], float) else w[k]) for k in ("value","ms_per_step","ms_per_step_same_graphs_no_input","shape_buckets","graphs_captured_inside_timed_region") if k in w} if "error" not in w else w)
PY
done
