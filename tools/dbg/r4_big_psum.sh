# S-BIG with / without the parked partial sums (run from the repo root on the GPU box)
mkdir -p gpurun_out
for i in 1 2; do
python bench.py --workload big --steps 30 --warmup 5 --no-live-pmc --no-sub --no-loop --no-cpu-baseline --no-stress 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('parked   ', j['value'], j['ms_per_step'], (j.get('parity') or {}).get('worst_max_abs_logit_err'))"
MOBGT_NO_PSUM_DEFER=1 python bench.py --workload big --steps 30 --warmup 5 --no-live-pmc --no-sub --no-loop --no-cpu-baseline --no-stress 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('immediate', j['value'], j['ms_per_step'])"
done
bash tools/prof_step.sh r4c_bench_big --workload big --steps 12 --warmup 4
grep -c "reduce_kernel" gpurun_out/r4c_bench_big_step_seq.txt; grep -n "partial_sum\|multi_tensor" gpurun_out/r4c_bench_big_step_seq.txt | cut -c1-120
