# S-BIG after the round-4 tail changes (run from the repo root on the GPU box)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "embed or skinny" 2>&1 | tail -4
for i in 1 2; do
python bench.py --workload big --steps 30 --warmup 5 --no-live-pmc --no-sub --no-loop --no-cpu-baseline --no-stress 2>gpurun_out/big_ab.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('big', j['value'], j['ms_per_step'])"
done
tail -3 gpurun_out/big_ab.err
bash tools/prof_step.sh r4b_bench_big --workload big --steps 12 --warmup 4
grep -n "scatter_add_runs\|skinny_bwd_both\|pack_mfma" gpurun_out/r4b_bench_big_step_seq.txt | cut -c1-120
