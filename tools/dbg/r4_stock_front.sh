# A/B of the stock front as one grid (run from the repo root on the GPU box)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_layer.py -q -x -k "stock" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "stock" 2>&1 | tail -3
for i in 1 2; do
python bench.py --variant stock --no-live-pmc --no-sub --no-loop --no-cpu-baseline --no-stress 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one grid', j['value'], j['ms_per_step'], j.get('launches_per_step'))"
MOBGT_NO_STOCK_FRONT=1 python bench.py --variant stock --no-live-pmc --no-sub --no-loop --no-cpu-baseline --no-stress 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('three   ', j['value'], j['ms_per_step'], j.get('launches_per_step'))"
done
