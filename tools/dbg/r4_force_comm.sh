# the S-FSQ step in its data-parallel form on ONE rank over RCCL (run from the repo root on the GPU box)
mkdir -p gpurun_out
for gc in fp32 bf16; do
timeout 600 python bench.py --force-comm --grad-comm $gc --no-cpu-baseline --no-stress > gpurun_out/r4_force_comm_$gc.json 2> gpurun_out/r4_force_comm_$gc.err
python - <<PY
import json
j = json.load(open("gpurun_out/r4_force_comm_$gc.json"))
print("$gc", round(j["value"], 1), round(j["ms_per_step"], 4), j["comm_backend"], j["rccl_ranks"], j["allreduce_exposed_us"], j["grad_comm_dtype"], j["forced_comm"], j["config"].get("parallelism"), j["parity"]["worst_max_abs_logit_err"])
PY
done
tail -3 gpurun_out/r4_force_comm_fp32.err
