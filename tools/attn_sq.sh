#!/bin/bash
# Developer tool (GPU box): SQ counters of the c5 attention kernels (one --pmc pass each set, --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  REPS=4 P=${P:-0.1} rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/sq_$i -o r -- python3 tools/attn_bwd_bench.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq_*/**/r_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_fwd" in n or "attn_bwd" in n:           # (not attn_dq_finish_kernel: it would halve the one-pass kernel's averages)
            k = "fwd" if "attn_fwd" in n else ("one" if "bwd_one" in n else ("dq" if "bwd_dq" in n else "dkv"))
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("fwd", "one", "dq", "dkv"):
    print(k, {c: round(sum(v[1:]) / max(len(v) - 1, 1) / 1e6, 3) for c, v in sorted(acc[k].items())})
PY
rm -rf gpurun_out/sq_*
