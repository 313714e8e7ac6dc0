mkdir -p gpurun_out
timeout 900 python tools/dbg/r4_onepass.py > gpurun_out/onepass.log 2>&1; echo "rc $?" >> gpurun_out/onepass.log
grep -v "^/opt\|Warn" gpurun_out/onepass.log | tail -18
ROUNDS=1 tools/r4_ab.sh > gpurun_out/ab4.log 2>&1
grep -v "^/opt" gpurun_out/ab4.log
