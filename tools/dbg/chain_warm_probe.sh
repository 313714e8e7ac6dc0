#!/bin/bash
# Developer tool (GPU box): stand-alone chain forward launches behind a 64 MB filler (cold) vs back to back (weights warm in every L2).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for mode in cold warm; do
  NOFILL=$([ $mode = warm ] && echo 1 || echo 0) rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cwp_$mode -o r -- python3 tools/chain_pmc.py 608 24 > /dev/null 2>&1
  python3 - $mode <<'PY'
import csv, glob, sys
mode = sys.argv[1]
ts = []
for f in glob.glob(f"gpurun_out/cwp_{mode}/**/r_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "layer_chain_fwd" in r["Kernel_Name"]:
            ts.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
ts = sorted(t for _, t in sorted(ts)[4:])
print(mode, "n", len(ts), "median", ts[len(ts) // 2], "min", ts[0], "max", ts[-1])
PY
  rm -rf gpurun_out/cwp_$mode
done
