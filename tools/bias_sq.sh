#!/bin/bash
# Developer tool (GPU box): SQ counters + HBM fetch of build_bias / build_bias_bwd at the c5-like shape (one --pmc pass per set).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  REPS=3 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/bsq_$i -o r -- python3 tools/bias_bwd_bench.py > /dev/null 2>&1
done
REPS=3 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bsq_t -o r -- python3 tools/bias_bwd_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/bsq_*/**/r_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "build_bias" in r["Kernel_Name"]:
            k = "bwd" if "bwd" in r["Kernel_Name"] else "fwd"
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("fwd", "bwd"):
    print(k, {c: round(sum(v[1:]) / max(len(v) - 1, 1) / 1e6, 3) for c, v in sorted(acc[k].items())})
for f in glob.glob("gpurun_out/bsq_t/**/r_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "build_bias" in r["Name"]:
            print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
rm -rf gpurun_out/bsq_*
