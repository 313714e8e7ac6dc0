mkdir -p gpurun_out
for mn in 131072 0 131072 0; do
MOBGT_WGRAD_HIP_MN=$mn python bench.py --workload big --steps 30 --warmup 5 --no-live-pmc --no-sub --no-loop --no-cpu-baseline --no-stress 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hip wgrad up to $mn outputs:', j['value'], j['ms_per_step'])"
done
