set -x
python bench.py > gpurun_out/r3_bench_fsq.json 2> gpurun_out/r3_bench_fsq.err

bash tools/prof_step.sh r3_bench_fsq > /dev/null
python bench.py --workload gow > gpurun_out/r3_bench_gow.json 2> gpurun_out/r3_bench_gow.err
python bench.py --variant stock --no-live-pmc > gpurun_out/r3_bench_stock.json 2> gpurun_out/r3_bench_stock.err
python bench.py --workload big --steps 30 --warmup 5 --no-live-pmc > gpurun_out/r3_bench_big.json 2> gpurun_out/r3_bench_big.err
bash tools/prof_step.sh r3_bench_big --workload big --steps 12 --warmup 4 > /dev/null
ls gpurun_out | head -50
