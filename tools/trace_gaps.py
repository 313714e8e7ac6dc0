"""Developer tool: gaps > 2 us between consecutive kernels of one training step in a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "build_bias_kernel" in r["Kernel_Name"]]
a, b = idx[-4], idx[-3]
for i in range(a, b):
    gap = (int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3
    if gap > 2.0:
        print(f"{gap:7.1f} us after {rows[i]['Kernel_Name'][:70]}  before {rows[i + 1]['Kernel_Name'][:60]}")
