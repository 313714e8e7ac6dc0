"""Where does the bf16 configuration's error on the REAL Gowalla batch (golden G8, batch A) come from?  Runs the fq Graphormer on
the device twice -- f32 configuration (within 8e-6 of the reference's logits, tests/test_gpu_real.py) and bf16 configuration (what
bench.py times) -- and prints the relative L2 / max error of every stage of the forward: GCN tables, bias, tokens, each layer's
output, head, logits.  No oracle involved: the f32 device run is the yardstick."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden"):
    sys.path.insert(0, p)
from inputs import real_universe, real_trajs                     # noqa: E402
from mobgt_amd.data import DeviceCollator, make_bin_table       # noqa: E402
from mobgt_amd.model_fqandtoyo import Graphormer                 # noqa: E402
from test_oracle_model import seeded_state                        # noqa: E402

z = np.load(ROOT + "/tests/golden/g8_gowalla_real.npz")
uni = real_universe(z)
nb, edges, table = make_bin_table(uni.distance)
VARIANTS = {
    "f32": {},
    "bf16": dict(bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16, act_dtype=torch.bfloat16),
    "bf16 act only": dict(act_dtype=torch.bfloat16),
    "bf16 bias only": dict(bias_dtype=torch.bfloat16),
    "bf16 gcn only": dict(gcn_dtype=torch.bfloat16),
}
only = sys.argv[1:] or list(VARIANTS)
stages = {}
for name in ["f32"] + [v for v in only if v != "f32"]:
    kw = VARIANTS[name]
    m = Graphormer(n_layers=6, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01,
                   ffn_dim=1024, dataset_name="gowalla_nevda", warmup_updates=10, tot_updates=100, peak_lr=2e-4, end_lr=1e-9,
                   edge_type="multi_hop", multi_hop_max_dist=20, attention_dropout_rate=0.1, universe=uni, **kw)
    names = [str(n) for n in z["param_names"]]
    shapes = [eval(str(s)) for s in z["param_shapes"]]
    sd = {k: v.detach() for k, v in seeded_state(list(zip(names, shapes)), int(z["seed"])).items()}
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    coll = DeviceCollator("cuda", bin_table=table)
    b = coll(real_trajs(z, "a"))
    rec = {}

    def hook(tag):
        def f(mod, inp, out):
            o = out[0] if isinstance(out, (tuple, list)) else out
            parts = getattr(o, "_mobgt_parts", ())
            o = o.detach().float()
            for p_ in parts:
                o = o + p_.detach().float()
            rec[tag] = o.cpu()
        return f
    m.poi_distance_model.register_forward_hook(hook("poidist (rows the batch reads)"))
    m.poi_cat_model.register_forward_hook(hook("catemb"))
    for li, l in enumerate(m.layers):
        l.register_forward_hook(hook(f"layer {li} out"))
    nf0, ab0 = m.node_features, m.assemble_bias

    def nf(*a, **k):
        o = nf0(*a, **k)
        rec["tokens x0"] = o.detach().float().cpu()
        return o

    def ab(*a, **k):
        o = ab0(*a, **k)
        d = o.dense().cpu()
        rec["bias (finite part)"] = torch.where(torch.isfinite(d), d, torch.zeros_like(d))
        return o
    m.node_features, m.assemble_bias = nf, ab
    with torch.no_grad():
        out = m(b)
    rec["logits"] = out[0].float().cpu()
    rec["golden logits"] = torch.from_numpy(z["a/logits"]).float()
    stages[name] = rec
    if name != "f32":
        print("==== %s vs f32 configuration (G8 batch A: N = 1, 2, 5, 17, 94, 8, 12, 30)" % name)
        for k, v in rec.items():
            w = stages["f32"][k]
            if v.shape != w.shape:
                print("%-34s shapes differ %s %s" % (k, tuple(v.shape), tuple(w.shape)))
                continue
            d = (v - w).double()
            print("%-34s relL2 %.5f  max|err| %.4e  rms(ref) %.4e  max|ref| %.4e" % (
                k, float(d.norm() / w.double().norm().clamp_min(1e-30)), float(d.abs().max()), float(w.double().pow(2).mean().sqrt()),
                float(w.abs().max())))
    else:
        print("f32 configuration vs golden logits: max|err| %.3e" % float((rec["logits"] - rec["golden logits"]).abs().max()))

# ---- inside layer 0: which rounding point carries the error?  (the fused node's saved tensors, f32 against bf16 configuration)
print("==== inside layer 0 (saved tensors of the fused node): bf16 against f32 configuration")
saved = {}
for name in ("f32", "bf16"):
    kw = VARIANTS[name]
    m = Graphormer(n_layers=6, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01,
                   ffn_dim=1024, dataset_name="gowalla_nevda", warmup_updates=10, tot_updates=100, peak_lr=2e-4, end_lr=1e-9,
                   edge_type="multi_hop", multi_hop_max_dist=20, attention_dropout_rate=0.1, universe=uni, **kw)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    b = DeviceCollator("cuda", bin_table=table)(real_trajs(z, "a"))
    from mobgt_amd import fused_layer
    fused_layer._CHAIN[0] = False              # separate launches: every intermediate is materialised
    from mobgt_amd.model import refresh_shadows
    pack = m.assemble_bias(b)
    x0 = stages["f32"]["tokens x0"].cuda().requires_grad_(True)       # the SAME input for both
    refresh_shadows(m.layers)
    out = m.layers[0](x0, pack)
    names = ("x", "xa", "qkv", "a", "lse", "x1", "z", "u", "h", "x2")
    saved[name] = {k: (None if t is None else t.detach().float().cpu()) for k, t in zip(names, out.grad_fn.saved_tensors)}
    saved[name]["out"] = out.detach().float().cpu()
    fused_layer._CHAIN[0] = True
for k, w in saved["f32"].items():
    v = saved["bf16"].get(k)
    if w is None or v is None or k == "lse" or v.shape != w.shape:
        continue
    d = (v - w).double()
    print("%-6s relL2 %.5f  max|err| %.4e  rms(ref) %.4e  max|ref| %.4e   row-mean rms %.4e" % (
        k, float(d.norm() / w.double().norm().clamp_min(1e-30)), float(d.abs().max()), float(w.double().pow(2).mean().sqrt()),
        float(w.abs().max()), float(w.double().mean(-1).pow(2).mean().sqrt())))
