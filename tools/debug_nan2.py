import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mobgt_amd import synth
from mobgt_amd.data import DeviceCollator, make_bin_table
from mobgt_amd.model_fqandtoyo import Graphormer
from mobgt_amd.train import TrainStep
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
graph = sys.argv[1] == "graph"
if os.environ.get('PREF_BLAS'):
    print('preferred blas ->', torch.backends.cuda.preferred_blas_library(os.environ['PREF_BLAS']))
dev = torch.device("cuda", 0)
uni = synth.make_universe(P=P, n_cat=300, n_user=1080, seed=1)
nb, _, table = make_bin_table(uni.distance)
coll = DeviceCollator(dev, bin_table=table)
batches = [coll(synth.make_batch_of_trajectories(seed=1000 + i, G=16, P=P, n_user=1080, cat_of_poi=uni.cat_of_poi)) for i in range(8)]
print("N per batch", [b.x.shape[1] for b in batches])
torch.manual_seed(1)
kw = dict(bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16, act_dtype=torch.bfloat16)
for k in os.environ.get('F32', '').split(','):
    if k: kw[k] = torch.float32
args = dict(bench.MODEL_ARGS)
if os.environ.get('DROPCFG'):
    a, b, c = (float(v) for v in os.environ['DROPCFG'].split(','))
    args.update(dropout_rate=a, intput_dropout_rate=b, attention_dropout_rate=c)
if os.environ.get('NODROP'):
    args.update(dropout_rate=0.0, intput_dropout_rate=0.0, attention_dropout_rate=0.0)
print({k: args[k] for k in ('dropout_rate','intput_dropout_rate','attention_dropout_rate')})
model = Graphormer(universe=uni, num_bins=nb + 2, **kw, **args).to(dev)
from mobgt_amd import ops
if os.environ.get('TRACE'):
    ops.nan_trace_enable(dev)
ts = TrainStep(model, batches, use_graph=graph)
ts.prepare()
for i in range(int(sys.argv[3]) if len(sys.argv) > 3 else 120):
    l = float(ts.step(i).item())
    if os.environ.get('TRACE'):
        gbad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        pbad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
        if gbad or pbad:
            print("step", i, "batch", i % 8, "loss", l, "non-finite grads:", gbad[:6], len(gbad), "params:", pbad[:4], len(pbad))
            print([n for n, f in ops.nan_trace_report() if f])
            for n, p in model.named_parameters():
                if p.grad is not None and not torch.isfinite(p.grad).all():
                    g = p.grad
                    print("   ", n, tuple(g.shape), "nan", int(torch.isnan(g).sum()), "inf", int(torch.isinf(g).sum()))
            break
    if l != l:
        print("first NaN at step", i, "batch", i % 8)
        if os.environ.get('TRACE'):
            print([n for n, f in ops.nan_trace_report() if f])
        bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        print("non-finite grads:", bad[:12])
        break
else:
    print("no NaN in 120 steps, last loss", l)
