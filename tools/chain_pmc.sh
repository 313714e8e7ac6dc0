#!/bin/bash
# Developer tool (GPU box): HBM traffic per launch of the chain forward kernel at R rows (default 608), cold L2, from rocprofv3
# PMC counters as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes with --kernel-trace
# only; values are KB; FETCH_SIZE doubled on gfx950, WRITE_SIZE exact.  Writes gpurun_out/chain_pmc.json + the raw CSVs.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=${1:-608}
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/cpmc_$C -o r -- python3 tools/chain_pmc.py $R 20 > /dev/null 2>&1
  f=$(find gpurun_out/cpmc_$C -name "r_counter_collection.csv" | head -1)
  grep -E "Counter_Name|layer_chain" "$f" > gpurun_out/chain_pmc_$C.csv
done
python3 - "$R" <<'PY'
import csv, json, sys
out = {"rows": int(sys.argv[1])}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"gpurun_out/chain_pmc_{c}.csv")) if r["Counter_Name"] == c]
    out[c + "_KB_raw"] = sum(v) / len(v)
    out["launches"] = len(v)
    out["kernel"] = next(csv.DictReader(open(f"gpurun_out/chain_pmc_{c}.csv")))["Kernel_Name"][:80]
out["fetch_bytes_corrected"] = int(out["FETCH_SIZE_KB_raw"] * 1024 * 2)
out["write_bytes"] = int(out["WRITE_SIZE_KB_raw"] * 1024)
out["traffic_bytes"] = out["fetch_bytes_corrected"] + out["write_bytes"]
json.dump(out, open("gpurun_out/chain_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/cpmc_*
