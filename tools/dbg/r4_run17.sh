mkdir -p gpurun_out
timeout 900 python tools/dbg/r4_distgcn_probe.py > gpurun_out/probe17.log 2>&1; echo rc $? >> gpurun_out/probe17.log
grep -v Warning gpurun_out/probe17.log | cut -c1-200 | tail -70
