"""Developer tool: summarise one training step out of a rocprofv3 kernel_trace.csv (kernels between two
consecutive build_bias launches): count, busy time, gaps, top kernels."""
import csv, sys, collections, glob
path = sys.argv[1] if len(sys.argv) > 1 else glob.glob("gpurun_out/prof*/runc/*_kernel_trace.csv")[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "step_prologue_kernel" in r["Kernel_Name"]]       # (the first launch of a step)
k = int(sys.argv[2]) if len(sys.argv) > 2 else -4
a, b = idx[k], idx[k + 1]
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"step wall {(t1 - t0) / 1e3:.1f} us, kernels {len(seg)}, busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
names = collections.OrderedDict()
for r in seg:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    names.setdefault(r["Kernel_Name"][:100], []).append(d)
for name, v in sorted(names.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{name:100s} n={len(v):4d} tot={sum(v):8.1f} avg={sum(v) / len(v):7.1f}")
