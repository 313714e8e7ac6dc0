"""Developer tool: build_bias / build_bias_bwd alone at the c5-like shape (for rocprofv3 --kernel-trace / --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobgt_amd import synth, ops
from mobgt_amd.data import DeviceCollator, make_bin_table
from mobgt_amd.model_fqandtoyo import Graphormer
P, N, G, L = 7856, int(os.environ.get("N", 784)), 16, int(os.environ.get("L", 12))
dev = torch.device("cuda", 0)
uni = synth.make_universe(P=P, n_cat=300, n_user=1080, seed=1)
nb, _, table = make_bin_table(uni.distance)
from mobgt_amd.workloads import FSQ_MODEL_ARGS
args = dict(FSQ_MODEL_ARGS); args.update(n_layers=1, hidden_dim=192, multi_hop_max_dist=int(os.environ.get("DH", 20)))
model = Graphormer(universe=uni, num_bins=nb + 2, bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16, act_dtype=torch.bfloat16, **args).to(dev)
coll = DeviceCollator(dev, bin_table=table)
trajs = synth.make_batch_of_trajectories(seed=5, G=G, P=P, n_user=1080, cat_of_poi=uni.cat_of_poi, n_nodes=[N] * G)
batch = coll(trajs)
for rep in range(int(os.environ.get("REPS", 3))):
    pack = model.assemble_bias(batch)
    pack.needs_grad = True
    pack.n_use = L
    buf = pack.grad_buffer()
    if rep == 0:
        src = torch.randn(buf.shape[1:], device=dev).bfloat16()
    for l in range(L):
        buf[l].copy_(src)
    pack.n_bwd = L
    pack.token.backward()
torch.cuda.synchronize()
print("ok", float(model.rel_pos_encoder.weight.grad.abs().sum()), float(model.edge_encoder.weight.grad.abs().sum()))
