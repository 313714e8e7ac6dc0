# one rank over RCCL: the data-parallel step host-issued (four replays + exchange calls) against ONE graph with captured exchanges
mkdir -p gpurun_out
for og in 0 1 0 1; do
for gc in fp32 bf16; do
MOBGT_DDP_ONE_GRAPH=$og timeout 600 python bench.py --force-comm --grad-comm $gc --no-cpu-baseline --no-stress > gpurun_out/r4_one_graph_${og}_$gc.json 2>/dev/null
python -c "import json; j=json.load(open('gpurun_out/r4_one_graph_${og}_$gc.json')); print('one_graph=$og $gc:', round(j['value'],1), round(j['ms_per_step'],4), j['ddp_one_graph'], j['comm_backend'], j['rccl_ranks'], j['allreduce_exposed_us'], j['parity']['worst_max_abs_logit_err'])"
done
done
