#!/bin/bash
# Developer tool (GPU box): kernel trace of the default bench run; prints one training step's kernel sequence.
#   tools/prof_step.sh <tag> [bench args]  -> gpurun_out/<tag>_step_seq.txt, gpurun_out/<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=${1:-step}; shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -o r -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-stress --no-live-pmc --no-loop --no-sub "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
f=$(find gpurun_out/prof_${tag} -name "r_kernel_trace.csv" | head -1)
python3 tools/trace_seq.py "$f" > gpurun_out/${tag}_step_seq.txt
python3 tools/trace_step.py "$f" -4 40 > gpurun_out/${tag}_step_summary.txt
python3 tools/filter_stats.py "$(find gpurun_out/prof_${tag} -name r_kernel_stats.csv | head -1)" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_${tag}
head -3 gpurun_out/${tag}_step_summary.txt
