#!/bin/bash
# Developer tool (GPU box): per-kernel durations (rocprofv3 --kernel-trace --stats, one warm input set) of the shipped attention
# library and of variant builds (tools/attn_ab.sh), then the interleaved all-cold A/B of tools/attn_ab.py.
#   tools/r4_ab.sh name1 name2 ...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in ship "$@"; do
  lib=$PWD/mobgt_amd/libmobgt_hip.so
  [ "$v" != ship ] && lib=$PWD/mobgt_amd/libmobgt_hip_ab_$v.so
  MOBGT_HIP_LIB=$lib REPS=10 P=0.1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$v -o r -- python3 tools/attn_bwd_bench.py > /dev/null 2>&1
  f=$(find gpurun_out/ab_$v -name "r_kernel_stats.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "attn" in r["Name"]:
        print("%-10s %-44s calls %s avg %.1f us" % (sys.argv[2], r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/ab_$v
done
ROUNDS=${ROUNDS:-3} python3 tools/attn_ab.py "$@"
