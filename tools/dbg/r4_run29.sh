mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_distgcn.py tests/test_gpu_kernels.py tests/test_gpu_bench_parity.py -x -q > gpurun_out/t29.log 2>&1; echo "pytest rc $?" >> gpurun_out/t29.log
tail -3 gpurun_out/t29.log
for v in 0 1; do
if [ $v = 1 ]; then export MOBGT_NO_SGEMM_SPLITK=1; fi
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-loop --no-live-pmc --no-sub --no-parity > gpurun_out/b29_$v.json 2> gpurun_out/b29_$v.err
python - <<PY
import json
j=json.load(open('gpurun_out/b29_$v.json')); print("fsq no_splitk=$v", j["value"], j["ms_per_step"])
PY
done
unset MOBGT_NO_SGEMM_SPLITK
bash tools/prof_step.sh r4h_fsq > /dev/null 2>&1
sed -n 4,6p gpurun_out/r4h_fsq_step_seq.txt | cut -c1-100
