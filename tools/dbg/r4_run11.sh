mkdir -p gpurun_out
ROUNDS=2 tools/r4_ab.sh skip1 skip2 skip3 skip4 > gpurun_out/ab3.log 2>&1
cat gpurun_out/ab3.log | grep -v "^/opt"
