"""Developer tool: torch.profiler table of one eager train step of the bench workload (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mobgt_amd import synth
from mobgt_amd.data import DeviceCollator, make_bin_table
from mobgt_amd.model_fqandtoyo import Graphormer
from mobgt_amd.train import TrainStep

P = int(os.environ.get("P", "7856"))
dev = torch.device("cuda", 0)
uni = synth.make_universe(P=P, n_cat=300, n_user=1080, seed=1)
nb, _, table = make_bin_table(uni.distance)
model = Graphormer(universe=uni, num_bins=nb + 2, bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16, act_dtype=torch.bfloat16, **__import__("mobgt_amd.workloads", fromlist=["x"]).FSQ_MODEL_ARGS).to(dev)
coll = DeviceCollator(dev, bin_table=table)
batches = [coll(synth.make_batch_of_trajectories(seed=1001 + i, G=16, P=P, n_user=1080, cat_of_poi=uni.cat_of_poi)) for i in range(2)]
ts = TrainStep(model, batches, autocast_dtype=None, use_graph=False)
for i in range(3):
    ts.step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    ts.step(1)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::sum", "aten::mean", "aten::cat", "aten::zeros", "aten::fill_", "aten::copy_", "aten::add", "aten::mul", "aten::index", "aten::embedding", "aten::where")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:45]:
    print(f"{e.key:14s} n={e.count:3d} cuda_us={e.device_time_total:9.1f} shapes={str(e.input_shapes)[:110]}")
