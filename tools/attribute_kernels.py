"""Developer tool: one EAGER training step under torch.profiler; prints every kernel in launch order with the chain of
CPU ops (autograd node > aten op) that launched it -- to find which Python line owns a stray copy / fill launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mobgt_amd import synth
from mobgt_amd.data import DeviceCollator, make_bin_table
from mobgt_amd.model_fqandtoyo import Graphormer
from mobgt_amd.train import TrainStep

dev = torch.device("cuda", 0)
torch.manual_seed(0)
uni = synth.make_universe(P=7856, n_cat=300, n_user=1080, seed=0)
num_bins, _, table = make_bin_table(uni.distance)
model = Graphormer(universe=uni, num_bins=num_bins + 2, bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16,
                   act_dtype=torch.bfloat16, fused_layers=True, **bench.MODEL_ARGS).to(dev)
coll = DeviceCollator(dev, bin_table=table, multi_hop_max_dist=20, rel_pos_max=1024)
batches = []
for i in range(2):
    trajs = synth.make_batch_of_trajectories(seed=1000 + i, G=16, P=7856, n_user=1080, cat_of_poi=uni.cat_of_poi, hi=256)
    batches.append(coll(trajs))
ts = TrainStep(model, batches, use_graph=False, seed=0)
ts.prepare()
for i in range(4):
    ts.step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    ts.step(4)
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if not ev.kernels:
        continue
    # only the innermost op that owns the launch
    if any(c.kernels for c in ev.cpu_children):
        continue
    chain, p = [], ev
    while p is not None:
        chain.append(p.name)
        p = p.cpu_parent
    for k in ev.kernels:
        rows.append((ev.time_range.start, k.name[:60], " < ".join(n[:48] for n in chain[:4])))
rows.sort()
for n, (t, k, c) in enumerate(rows):
    print(f"{n:4d} {k:60s} | {c}")
