"""Developer tool: c5-like stress step (G=16 graphs x N=784 nodes, C=256/d=32, 12 layers) -- where does the time go."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mobgt_amd import synth, ops
from mobgt_amd.data import DeviceCollator, make_bin_table
from mobgt_amd.model_fqandtoyo import Graphormer
from mobgt_amd.train import TrainStep
P, N, G, L = 7856, int(os.environ.get("N", 784)), 16, int(os.environ.get("L", 12))
dev = torch.device("cuda", 0)
uni = synth.make_universe(P=P, n_cat=300, n_user=1080, seed=1)
nb, _, table = make_bin_table(uni.distance)
args = dict(bench.MODEL_ARGS); args.update(n_layers=L, hidden_dim=192)
model = Graphormer(universe=uni, num_bins=nb + 2, bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16, act_dtype=torch.bfloat16, **args).to(dev)
coll = DeviceCollator(dev, bin_table=table)
t0 = time.perf_counter()
trajs = synth.make_batch_of_trajectories(seed=5, G=G, P=P, n_user=1080, cat_of_poi=uni.cat_of_poi, n_nodes=[N] * G)
t1 = time.perf_counter()
batch = coll(trajs); torch.cuda.synchronize()
t2 = time.perf_counter()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
h = coll.pack_host(trajs)
d = {k: torch.from_numpy(v).to(dev) for k, v in h.items()}
torch.cuda.synchronize(); e0.record(); b2 = coll.finish(d); e1.record(); torch.cuda.synchronize()
print(f"synth {t1-t0:.2f}s, collate(host pack + H2D + device) {t2-t1:.3f}s, device part {e0.elapsed_time(e1):.2f} ms for {G} graphs of {N} nodes")
ts = TrainStep(model, [batch], use_graph=bool(int(os.environ.get('GRAPH', '0'))))
ts.prepare()
for i in range(3): ts.step(0)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(5):
    ta = time.perf_counter(); ts.step(0); tb = time.perf_counter(); torch.cuda.synchronize(); tc = time.perf_counter()
    print(f"  step {i}: launch {1e3*(tb-ta):.2f} ms, drain {1e3*(tc-tb):.2f} ms, reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB")
dt = (time.perf_counter() - t) / 5
print(f"train step {dt*1e3:.2f} ms -> {G/dt:.1f} check-ins/s")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    ts.step(0); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:int(os.environ.get("TOP", 16))]
for e in rows:
    print(f"{e.key[:90]:90s} n={e.count:4d} total_us={e.device_time_total:10.1f}")
