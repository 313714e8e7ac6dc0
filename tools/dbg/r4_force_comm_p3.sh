mkdir -p gpurun_out
for i in 1 2 3 4; do
MOBGT_DDP_PARTS=3 timeout 600 python bench.py --force-comm --no-cpu-baseline --no-stress --no-parity > gpurun_out/fc3_$i.json 2> gpurun_out/fc3_$i.err
echo "run $i rc=$? lines=$(wc -l < gpurun_out/fc3_$i.json)"
tail -3 gpurun_out/fc3_$i.err | cut -c1-300
done
