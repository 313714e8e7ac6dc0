"""Developer tool (GPU box): one bucket's step graph of train.EpochLoop under rocprofv3 --kernel-trace --stats: what the
device collate adds to the replayed step.  python tools/loop_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobgt_amd import workloads
from mobgt_amd.train import EpochLoop
uni, model, coll = workloads.build("fsq", "cuda", seed=1)
pool = workloads.make_pool("fsq", 64, 16, uni, seed0=5000)
data = [t for trajs in pool for t in trajs]
loop = EpochLoop(model, coll, data, batch_size=16, seed=1)
loop.run_epoch(0); loop.run_epoch(1)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
for ep in range(2, 2 + int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    n += loop.run_epoch(ep)["steps"]
torch.cuda.synchronize()
print("ms/step %.4f over %d steps" % ((time.perf_counter() - t0) / n * 1e3, n))
# where does the host spend its time?
import time as _t
acc = {"stage": 0.0, "launch": 0.0, "pack": 0.0, "check": 0.0}
_stage, _launch, _pack, _check = loop._stage, loop._launch, coll.pack_host, loop._check_host
def stage(ids):
    t = _t.perf_counter(); r = _stage(ids); acc["stage"] += _t.perf_counter() - t; return r
def launch(slot, st):
    t = _t.perf_counter(); r = _launch(slot, st); acc["launch"] += _t.perf_counter() - t; return r
def pack(*a, **k):
    t = _t.perf_counter(); r = _pack(*a, **k); acc["pack"] += _t.perf_counter() - t; return r
def check(h):
    t = _t.perf_counter(); r = _check(h); acc["check"] += _t.perf_counter() - t; return r
_fin = coll.finish_into
acc["finish_into"] = 0.0
def fin(*a, **k):
    t = _t.perf_counter(); r = _fin(*a, **k); acc["finish_into"] += _t.perf_counter() - t; return r
coll.finish_into = fin
_step = loop.ts.step
acc["ts.step"] = 0.0
def step(i):
    t = _t.perf_counter(); r = _step(i); acc["ts.step"] += _t.perf_counter() - t; return r
loop.ts.step = step
loop._stage, loop._launch, coll.pack_host, loop._check_host = stage, launch, pack, check
n = 0
for ep in range(20, 24):
    n += loop.run_epoch(ep)["steps"]
torch.cuda.synchronize()
print("host us/step:", {k: round(v / n * 1e6, 1) for k, v in acc.items()})
