"""Developer tool (GPU box, after tools/attn_stamp.sh): where a wave of the attention forward spends its cycles at the c5
shape.  MOBGT_HIP_LIB must point at the stamp build; prints the share of each phase of the chunk loop (median over waves)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MOBGT_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mobgt_amd", "libmobgt_hip_stamp.so"))
import torch
from mobgt_amd import ops
G, H, T, d = 16, 8, 785, 32
C = H * d
g = torch.Generator().manual_seed(0)
sets = []
for _ in range(5):
    qkv = torch.randn(G, T, 3 * C, generator=g).cuda().bfloat16()
    bias = torch.randn(G, H, T, T, generator=g).cuda()
    pack = ops.pack_bias(bias, G, H, T, dtype=torch.bfloat16)
    del bias
    sets.append((qkv, pack))
names = ["bias wait+park", "(fine build: LDS bias + QK^T until S is available, both tiles)", "K/V wait+store", "barrier", "tile 0 (fine: rest)", "tile 1 (fine: rest)", "-", "-"]
for rep in range(3):
    for qkv, pack in sets:
        out, lse = ops._attn_fwd(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], pack, d ** -0.5, 0.1, 1, None)
torch.cuda.synchronize()
l = lse.view(G * H, T)
rows = torch.cat([l[:, q0:q0 + 8] for q0 in range(0, T - 8, 32)], 0)        # one row of 8 sums per wave
tot = rows[:, :6].sum(1)
print("waves", rows.shape[0], "median total cycles %.0f  (min %.0f max %.0f)" % (tot.median(), tot.min(), tot.max()))
life = rows[:, 7].median() * 0.01
print("median wave lifetime %.1f us  -> shader clock %.2f GHz; longest wave %.1f us" % (life, float(tot.median()) / float(life) / 1e3, float(rows[:, 7].max()) * 0.01))
st = rows[:, 6]
st = (st - st.min()) * 0.01
end = st + rows[:, 7] * 0.01
import numpy as np
h, edges = np.histogram(st.cpu().numpy(), bins=12)
print("wave START times (us after the first): ", " ".join("%.0f:%d" % (edges[i], h[i]) for i in range(len(h))))
print("last wave ends %.1f us after the first starts; waves starting later than 5 us: %d" % (float(end.max()), int((st > 5).sum())))
for k in range(6):
    print("%-16s median %8.0f cycles  %5.1f %%" % (names[k], rows[:, k].median(), 100 * float(rows[:, k].median() / tot.median())))
# what makes a wave slow?
import numpy as np
stn, lifen = st.cpu().numpy(), (rows[:, 7] * 0.01).cpu().numpy()
print("corr(start time, lifetime) = %.2f; corr(start, end) = %.2f" % (np.corrcoef(stn, lifen)[0, 1], np.corrcoef(stn, stn + lifen)[0, 1]))
order = np.argsort(stn)
for lo, hi in ((0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)):
    sel = order[int(lo * len(order)):int(hi * len(order))]
    print("start quartile %.2f-%.2f: start %.1f us, lifetime %.1f us, end %.1f us" % (lo, hi, stn[sel].mean(), lifen[sel].mean(), (stn[sel] + lifen[sel]).mean()))

