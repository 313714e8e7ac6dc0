#!/bin/bash
# Developer tool (GPU box): per-kernel durations of the c5-shape attention forward / backward under rocprofv3.
#   tools/prof_attn.sh <tag>   -> gpurun_out/<tag>_p<P>_kernel_stats.csv, printed summary
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=${1:-attn}
for P in ${PS:-0.1 0.0}; do
  REPS=10 P=$P rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$P -o r -- python3 tools/attn_bwd_bench.py > /dev/null 2>&1
  f=$(find gpurun_out/prof_${tag}_$P -name "r_kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/${tag}_p${P}_kernel_stats.csv
  python3 - "$f" "$P" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "attn" in r["Name"]:
        print("p=%s %-44s calls %s avg %.1f us" % (sys.argv[2], r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
