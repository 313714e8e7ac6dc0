#!/bin/bash
# Developer tool (GPU box): build_bias forward with parts compiled out (libmobgt_hip_v1/2/3.so: no hop gathers / no
# transposed copy / no row-major store) to see where its time goes.  Results are wrong by construction.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in base v1 v2 v3; do
  if [ $v = base ]; then unset MOBGT_HIP_LIB; else export MOBGT_HIP_LIB=$GRAFT_REPO_ROOT/mobgt_amd/libmobgt_hip_$v.so; fi
  REPS=3 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bv_$v -o r -- python3 tools/bias_bwd_bench.py > /dev/null 2>&1
  python3 - "$v" <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/bv_%s/**/r_kernel_stats.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if "build_bias_kernel" in r["Name"]:
            print(sys.argv[1], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/bv_$v
done
