#!/bin/bash
# Developer tool (GPU box): where a replayed step idles (gaps between consecutive kernels).  tools/prof_gaps.sh [bench args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gaps -o r -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-stress "$@" > /dev/null 2>&1
python3 tools/trace_gaps.py "$(find gpurun_out/prof_gaps -name r_kernel_trace.csv | head -1)"
rm -rf gpurun_out/prof_gaps
