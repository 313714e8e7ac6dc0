"""Developer tool: reduce a rocprofv3 `*_kernel_stats.csv` to what profiles/ keeps -- every kernel of this repo, then the
50 most expensive library / torch kernels with >= 100 calls (the TunableOp trial kernels of the warm-up are left out)."""
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
ours = ("GLOBAL__N_", "anonymous namespace")
mine = [r for r in rows if any(t in r["Name"] for t in ours) and "at::native" not in r["Name"]]
other = [r for r in rows if r not in mine and int(r["Calls"]) >= 100]
other.sort(key=lambda r: -int(r["TotalDurationNs"]))
mine.sort(key=lambda r: -int(r["TotalDurationNs"]))
with open(dst, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    for r in mine + other[:50]:
        w.writerow(r)
print(len(mine), "own kernels,", min(50, len(other)), "library kernels")
