"""Poison every torch.empty/empty_like (NaN fill) to expose reads of uninitialised memory in eager mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
_e, _el = torch.empty, torch.empty_like
def empty(*a, **k):
    t = _e(*a, **k)
    return t.fill_(float("nan")) if t.is_floating_point() else t
def empty_like(*a, **k):
    t = _el(*a, **k)
    return t.fill_(float("nan")) if t.is_floating_point() else t
torch.empty, torch.empty_like = empty, empty_like
from mobgt_amd import synth
from mobgt_amd.data import DeviceCollator, make_bin_table
from mobgt_amd.model_fqandtoyo import Graphormer
from mobgt_amd.train import TrainStep
P = 2000
dev = torch.device("cuda", 0)
uni = synth.make_universe(P=P, n_cat=300, n_user=1080, seed=1)
nb, _, table = make_bin_table(uni.distance)
coll = DeviceCollator(dev, bin_table=table)
batches = [coll(synth.make_batch_of_trajectories(seed=1000 + i, G=16, P=P, n_user=1080, cat_of_poi=uni.cat_of_poi)) for i in range(8)]
for act in (torch.float32, torch.bfloat16):
    torch.manual_seed(1)
    model = Graphormer(universe=uni, num_bins=nb + 2, bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16, act_dtype=act, **bench.MODEL_ARGS).to(dev)
    ts = TrainStep(model, batches, use_graph=False)
    ls = [float(ts.step(i).item()) for i in range(10)]
    print(act, ["%.4f" % v for v in ls])
    if any(v != v for v in ls):
        # locate: eval-mode forward pieces
        b = batches[0]
        model.eval()
        with torch.no_grad():
            bias = model.assemble_bias(b)
            print(" bias finite:", torch.isfinite(bias.bias[..., :bias.T][torch.isfinite(bias.bias[..., :bias.T])]).all().item(), "nan:", torch.isnan(bias.bias).any().item())
            nf = model.node_features(b)
            print(" node_features nan:", torch.isnan(nf).any().item())
            out = nf
            for li, layer in enumerate(model.layers):
                out = layer(out, bias)
                print("  layer", li, "nan:", torch.isnan(out).any().item())
        model.train()
