#!/bin/bash
# Developer tool (GPU box): tools/dbg/bias_fwd_probe.py under rocprofv3 --kernel-trace; prints the build_bias durations.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bfp -o r -- python3 tools/dbg/bias_fwd_probe.py > gpurun_out/bias_fwd_probe.log 2>&1
grep -v "^W2\|rocprof" gpurun_out/bias_fwd_probe.log | tail -12
python3 - <<'PY'
import csv, glob, os
ts = []
for f in glob.glob("gpurun_out/bfp/**/r_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "build_bias" in r["Kernel_Name"]:
            ts.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
ts = [t for _, t in sorted(ts)]
print("build_bias launches (us):", ts)
PY
rm -rf gpurun_out/bfp
