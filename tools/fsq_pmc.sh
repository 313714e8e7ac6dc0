#!/bin/bash
# Developer tool (GPU box): HBM traffic per launch of the S-FSQ step's attention forward and chain kernels, from rocprofv3
# PMC counters as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes with --kernel-trace
# only; values are KB; FETCH_SIZE doubled on gfx950 (16-byte-per-lane reads), WRITE_SIZE exact.  Averages over every
# launch of the (eager, not graph-replayed) steps of `bench.py --no-graph`.  Writes gpurun_out/fsq_pmc.json.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/fpmc_$C -o r -- python3 bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-live-pmc --no-loop ${BENCH_ARGS} > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"gpurun_out/fpmc_{c}/**/r_counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            n = r["Kernel_Name"]
            k = "attn_fwd" if "attn_fwd_kernel" in n else ("chain_fwd" if "layer_chain_fwd" in n else ("chain_bwd" if "layer_chain_bwd" in n else None))
            if k:
                acc[k].append(float(r["Counter_Value"]))
    raw[c] = {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
out = {}
for k in ("attn_fwd", "chain_fwd", "chain_bwd"):
    if k in raw["FETCH_SIZE"] and k in raw["WRITE_SIZE"]:
        f, nf = raw["FETCH_SIZE"][k]; w, nw = raw["WRITE_SIZE"][k]
        out[k] = dict(FETCH_SIZE_KB_raw=f, WRITE_SIZE_KB_raw=w, launches=[nf, nw], fetch_bytes_corrected=int(f * 1024 * 2),
                      write_bytes=int(w * 1024), traffic_bytes=int(f * 1024 * 2 + w * 1024))
json.dump(out, open("gpurun_out/fsq_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/fpmc_*
