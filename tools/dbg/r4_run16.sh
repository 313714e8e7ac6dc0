mkdir -p gpurun_out
for v in 0 1; do
MOBGT_NO_DIST_GCN_FUSED=$v timeout 900 python -m pytest "tests/test_gpu_bench_parity.py::test_eager_eval_logits_loss_and_elementwise_gradients_vs_oracle" -q -s -k fsq > gpurun_out/t16_$v.log 2>&1; echo "pytest rc $?" >> gpurun_out/t16_$v.log
grep -n "rel_pos_encoder\|poi_pos_encoder\|edge_dis_encoder\|time_embed_model_48\|pytest rc\|passed\|failed" gpurun_out/t16_$v.log | cut -c1-150
done
timeout 600 python -m pytest tests/test_gpu_distgcn.py -x -q > gpurun_out/t16_d.log 2>&1; tail -2 gpurun_out/t16_d.log
bash tools/prof_step.sh r4e_fsq > gpurun_out/prof16.log 2>&1
grep -n "mask_\|sgemm" gpurun_out/r4e_fsq_step_seq.txt | cut -c1-100
python - <<PY
import json
j=json.load(open('gpurun_out/r4e_fsq_bench.json')); print("fsq under profiler", j["value"], j["ms_per_step"])
PY
