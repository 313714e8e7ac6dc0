"""Developer tool: 2-rank (gloo, shared GPU) loss / gradient trajectory of TrainStep -- torchrun --nproc-per-node 2 tools/dbg_ddp.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
import bench
from mobgt_amd import synth
from mobgt_amd.data import DeviceCollator, make_bin_table
from mobgt_amd.model_fqandtoyo import Graphormer
from mobgt_amd.train import TrainStep, broadcast_parameters
dev = torch.device("cuda", 0)
PP = int(os.environ.get("PP", 600)); uni = synth.make_universe(P=PP, n_cat=int(os.environ.get("NCAT", 12)), n_user=1080, seed=3)
nb, _, table = make_bin_table(uni.distance)
args = dict(bench.MODEL_ARGS); args.update(n_layers=int(os.environ.get("L", 2)))
torch.manual_seed(0)
model = Graphormer(universe=uni, num_bins=nb + 2, bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16, act_dtype=torch.bfloat16, **args).to(dev)
broadcast_parameters(model)
coll = DeviceCollator(dev, bin_table=table)
batches = [coll(synth.make_batch_of_trajectories(seed=100 * rank + i, G=int(os.environ.get("GG", 4)), P=PP, n_user=1080, cat_of_poi=uni.cat_of_poi)) for i in range(int(os.environ.get("NB", 2)))]
ts = TrainStep(model, batches, use_graph=os.environ.get("GRAPH", "1") == "1", overlap=os.environ.get("OVERLAP", "1") == "1")
ts.prepare()
mode = os.environ.get('DBG', '')
if mode == 'noar': ts.flat.all_reduce_mean = lambda: None
if mode == 'nodiv':
    def ar():
        dist.all_reduce(ts.flat.flat)
    ts.flat.all_reduce_mean = ar
if mode == 'cpuar':
    def ar2():
        t = ts.flat.flat.cpu(); dist.all_reduce(t); ts.flat.flat.copy_(t.to(dev) / 2)
    ts.flat.all_reduce_mean = ar2
for i in range(5):
    l = float(ts.step(i))
    g = ts.flat.flat
    p = ts.flat_params.tensor
    with torch.no_grad():
        model.eval(); lg = model(batches[(i + 1) % len(batches)])[0].float(); model.train()
    extra = f" next-batch eval logits absmax {float(lg.abs().max()):.3f} mean {float(lg.mean()):.4f} nan {bool(torch.isnan(lg).any())}"
    print(f"rank {rank} step {i} loss {l:.6f} |g| {float(g.norm()):.4e} g finite {bool(torch.isfinite(g).all())} |p| {float(p.detach().norm()):.6f} lr {ts.lr:.3e}" + extra, flush=True)
dist.barrier()
