#!/bin/bash
# Developer tool (GPU box): durations of build_bias / build_bias_bwd at the c5-like shape and at the S-FSQ bench shape.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
REPS=3 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bsq_t -o r -- python3 tools/bias_bwd_bench.py > gpurun_out/bias_time.log 2>&1
tail -1 gpurun_out/bias_time.log
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/bsq_t/**/r_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "build_bias" in r["Name"]:
            print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
rm -rf gpurun_out/bsq_t
