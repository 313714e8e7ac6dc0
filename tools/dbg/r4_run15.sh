mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_distgcn.py tests/test_gpu_train_parity.py tests/test_gpu_bench_parity.py tests/test_gpu_train.py tests/test_gpu_model.py -x -q > gpurun_out/t15.log 2>&1; echo "pytest rc $?" >> gpurun_out/t15.log
tail -8 gpurun_out/t15.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub > gpurun_out/b15_fsq.json 2> gpurun_out/b15_fsq.err
MOBGT_NO_DIST_GCN_FUSED=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub > gpurun_out/b15_fsq_old.json 2> gpurun_out/b15_fsq_old.err
python bench.py --workload gow --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub > gpurun_out/b15_gow.json 2> gpurun_out/b15_gow.err
python - <<PY
import json
for n in ("fsq","fsq_old","gow"):
    try:
        j=json.load(open('gpurun_out/b15_%s.json'%n))
        print(n, j["value"], j["ms_per_step"], j.get("kernels_per_step"), j["parity"]["worst_max_abs_logit_err"] if j.get("parity") else None)
    except Exception as e:
        print(n, "failed", e)
PY
bash tools/prof_step.sh r4d_fsq > gpurun_out/prof15.log 2>&1
tail -3 gpurun_out/prof15.log
