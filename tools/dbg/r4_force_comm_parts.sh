# one rank over RCCL: the step's data-parallel form with 1 / 2 / 3 parts of the backward pass, and without the exchange calls
mkdir -p gpurun_out
for parts in 1 2 3 4; do
MOBGT_DDP_PARTS=$parts timeout 600 python bench.py --force-comm --no-cpu-baseline --no-stress --no-parity 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('parts $parts:', round(j['value'],1), round(j['ms_per_step'],4), 'exposed exchange calls', round(j['allreduce_exposed_us'],1), 'us')"
done
