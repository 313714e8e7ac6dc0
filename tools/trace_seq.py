"""Developer tool: print the kernel sequence of one training step out of a rocprofv3 kernel_trace.csv."""
import csv, sys, glob
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "step_prologue_kernel" in r["Kernel_Name"]]       # (the first launch of a step)
k = int(sys.argv[2]) if len(sys.argv) > 2 else -4
a, b = idx[k], idx[k + 1]
# rotate so the step starts at its first kernel (the one after adamw)
for n, r in enumerate(rows[a:b]):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("at::native::", "")
    print(f"{n:4d} {d:7.1f} {name[:110]}")
