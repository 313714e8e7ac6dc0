mkdir -p gpurun_out
timeout 900 python tools/dbg/r4_onepass.py > gpurun_out/onepass.log 2>&1; echo "rc $?" >> gpurun_out/onepass.log
grep -v "^/opt\|Warn" gpurun_out/onepass.log | tail -22
