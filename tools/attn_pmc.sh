#!/bin/bash
# Developer tool (GPU box): HBM traffic of the c5-shape attention kernels from rocprofv3 PMC counters, as
# MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (with --kernel-trace only), values
# are KB, FETCH_SIZE doubled on gfx950 for 16-B/lane streaming reads, WRITE_SIZE exact.  Writes gpurun_out/attn_pmc_r2.json.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  REPS=5 P=0.1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_$C -o r -- python3 tools/attn_bwd_bench.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/r_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "attn" in r["Kernel_Name"]:
            k = "fwd" if "attn_fwd" in r["Kernel_Name"] else ("dq" if "bwd_dq" in r["Kernel_Name"] else "dkv")
            acc[k].append(float(r["Counter_Value"]))
    raw[c] = {k: sum(v[1:]) / max(len(v) - 1, 1) for k, v in acc.items()}       # skip the first (cold) launch
G, H, T, C, s = 16, 8, 785, 256, 2
alg = {"fwd": G * (4 * T * C * s + H * T * T * s + H * T * 4),
       "dq": G * (6 * T * C * s + H * T * T * (s + s) + H * T * 8),
       "dkv": G * (7 * T * C * s + H * T * T * s + H * T * 8)}
res = {}
for k in ("fwd", "dq", "dkv"):
    fetch = raw["FETCH_SIZE"][k] * 1024 * 2
    write = raw["WRITE_SIZE"][k] * 1024
    res[k] = dict(FETCH_SIZE_KB_raw=raw["FETCH_SIZE"][k], WRITE_SIZE_KB_raw=raw["WRITE_SIZE"][k], fetch_bytes_corrected=int(fetch),
                  write_bytes=int(write), traffic_bytes=int(fetch + write), algorithmic_bytes=alg[k])
out["c5_fwd_drop_bf16"] = res["fwd"]
out["c5_bwd_dq_drop_bf16"] = res["dq"]
out["c5_bwd_dkv_drop_bf16"] = res["dkv"]
out["c5_bwd_drop_bf16"] = dict(traffic_bytes=res["dq"]["traffic_bytes"] + res["dkv"]["traffic_bytes"],
                               algorithmic_bytes=alg["dq"] + alg["dkv"])
json.dump(out, open("gpurun_out/attn_pmc_r2.json", "w"), indent=1)
for k, v in out.items():
    print(k, v["traffic_bytes"] / 1e6, "MB vs algorithmic", v["algorithmic_bytes"] / 1e6)
PY
for C in FETCH_SIZE WRITE_SIZE; do cp $(find gpurun_out/pmc_$C -name r_counter_collection.csv | head -1) gpurun_out/r2_attn_c5_pmc_$C.csv; done
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
