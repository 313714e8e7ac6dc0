"""Developer tool (GPU box): interleaved A/B of attention-kernel builds in ONE process on ONE device (box-to-box spread is
larger than the effects looked for).  Variants: mobgt_amd/libmobgt_hip_ab_<name>.so (tools/attn_ab.sh) + "ship"."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mobgt_amd import _lib
import bench

names = ["ship"] + sys.argv[1:]
handles = {}
for n in names:
    path = _lib.LIB_PATH if n == "ship" else os.path.join(ROOT, "mobgt_amd", f"libmobgt_hip_ab_{n}.so")
    h = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    handles[n] = h
_lib.lib()
P = float(os.environ.get("P", 0.1))
res = {n: [] for n in names}
for rnd in range(int(os.environ.get("ROUNDS", 3))):
    for n in names:
        _lib._lib = handles[n]
        f, b, _ = bench.time_attention(16, 8, 785, 32, torch.bfloat16, torch.bfloat16, reps=24, p_drop=P, backward=True)
        res[n].append((f * 1e6, b * 1e6))
for n in names:
    fs, bs = sorted(x[0] for x in res[n]), sorted(x[1] for x in res[n])
    print("%-12s fwd min %.1f med %.1f | bwd min %.1f med %.1f   (us, p=%.2f, rotating sets)" % (n, fs[0], fs[len(fs) // 2], bs[0], bs[len(bs) // 2], P))
