#!/bin/bash
# Developer tool (GPU box): A/B of two builds of the library on the c5-shape attention kernels, interleaved (box-to-box and
# run-to-run spread is larger than the effects looked for).  mobgt_amd/libmobgt_hip_base.so = the baseline build.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for which in base new; do
  if [ $which = base ]; then export MOBGT_HIP_LIB=$GRAFT_REPO_ROOT/mobgt_amd/libmobgt_hip_base.so; else unset MOBGT_HIP_LIB; fi
  REPS=10 P=0.1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$which -o r -- python3 tools/attn_bwd_bench.py > /dev/null 2>&1
  f=$(find gpurun_out/ab_$which -name "r_kernel_stats.csv" | head -1)
  python3 - "$f" "$which" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "attn" in n:
        k = "fwd" if "attn_fwd" in n else ("dq" if "bwd_dq" in n else "dkv")
        out.append("%s %.1f" % (k, float(r["AverageNs"]) / 1e3))
print(sys.argv[2], " ".join(sorted(out)))
PY
  rm -rf gpurun_out/ab_$which
done
done
