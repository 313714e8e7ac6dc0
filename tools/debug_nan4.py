"""Keep references to intermediates during graph capture; after each replay report the first NaN tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mobgt_amd import synth, ops
from mobgt_amd.data import DeviceCollator, make_bin_table
from mobgt_amd.model_fqandtoyo import Graphormer
from mobgt_amd.train import TrainStep
P = int(sys.argv[1]) if len(sys.argv) > 1 else 7856
dev = torch.device("cuda", 0)
uni = synth.make_universe(P=P, n_cat=300, n_user=1080, seed=1)
nb, _, table = make_bin_table(uni.distance)
coll = DeviceCollator(dev, bin_table=table)
batches = [coll(synth.make_batch_of_trajectories(seed=1000 + i, G=16, P=P, n_user=1080, cat_of_poi=uni.cat_of_poi)) for i in range(8)]
torch.manual_seed(1)
model = Graphormer(universe=uni, num_bins=nb + 2, bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16, act_dtype=torch.bfloat16, **bench.MODEL_ARGS).to(dev)
keep = {}
cur = [None]
ONLY = os.environ.get("KEEP", "").split(",")
def rec(name, t):
    if cur[0] is not None and any(name.startswith(k) for k in ONLY if k):
        keep.setdefault(cur[0], []).append((name, t))
    return t
orig_ab, orig_nf = model.assemble_bias, model.node_features
model.assemble_bias = lambda b: (lambda p: (rec("bias", p.bias), p)[1])(orig_ab(b))
model.node_features = lambda b: rec("node_features", orig_nf(b))
for li, layer in enumerate(model.layers):
    layer.register_forward_hook(lambda m, i, o, li=li: rec(f"layer{li}", o))
model.out_proj.register_forward_hook(lambda m, i, o: rec("logits", o))
model.poi_distance_model.register_forward_hook(lambda m, i, o: rec("poidist", o))
model.embed_fuse_model4.register_forward_hook(lambda m, i, o: rec("fuse4", o))
ts = TrainStep(model, batches, use_graph=True)
orig_capture = ts._capture
def cap(i):
    with ts._on_stream():
        ts._fwd_bwd(batches[i])
    ts._join()
    g = torch.cuda.CUDAGraph()
    cur[0] = i
    with torch.cuda.graph(g, stream=ts.stream):
        ts._fwd_bwd(batches[i])
    cur[0] = None
    return g
ts._capture = cap
ts.prepare()
for i in range(60):
    l = float(ts.step(i).item())
    if l != l:
        gi = i % 8
        print("NaN at step", i, "graph", gi)
        for name, t in keep.get(gi, []):
            tt = t if t.dtype != torch.bfloat16 else t.float()
            fin = tt[torch.isfinite(tt) | torch.isnan(tt)] if name == "bias" else tt
            print(f"  {name:14s} nan={torch.isnan(fin).any().item()} max={fin[~torch.isnan(fin)].abs().max().item() if (~torch.isnan(fin)).any() else None}")
        break
else:
    print("no NaN")
