import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mobgt_amd import synth
from mobgt_amd.data import DeviceCollator, make_bin_table
from mobgt_amd.model_fqandtoyo import Graphormer
from mobgt_amd.train import TrainStep
P = 2000
dev = torch.device("cuda", 0)
uni = synth.make_universe(P=P, n_cat=300, n_user=1080, seed=1)
nb, _, table = make_bin_table(uni.distance)
coll = DeviceCollator(dev, bin_table=table)
batches = [coll(synth.make_batch_of_trajectories(seed=1001 + i, G=16, P=P, n_user=1080, cat_of_poi=uni.cat_of_poi)) for i in range(3)]
cfgs = {'a': (torch.float32, False), 'b': (torch.bfloat16, False), 'c': (torch.float32, True), 'd': (torch.bfloat16, True)}
for act, graph in [cfgs[k] for k in sys.argv[1]]:
    torch.manual_seed(1)
    model = Graphormer(universe=uni, num_bins=nb + 2, bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16, act_dtype=act, **bench.MODEL_ARGS).to(dev)
    ts = TrainStep(model, batches, use_graph=graph)
    ts.prepare()
    out = []
    for i in range(12):
        out.append(float(ts.step(i).item()))
    print(act, graph, ["%.4f" % v for v in out])
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    print("non-finite params:", bad[:8])
