import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mobgt_amd import ops, workloads
uni, model, coll = workloads.build("fsq", "cuda", seed=1, model_overrides=dict(n_layers=2))
batch = coll(workloads.make_pool("fsq", 1, 16, uni)[0])
model.train()
real = ops.take_bias_bwd_job
def spy():
    job = ops._BIAS_BWD_JOB.get("cur")
    print("job present:", job is not None)
    if job is not None:
        pack = job["pack"]()
        print("pack alive", pack is not None)
        if pack is not None:
            print("dbias", pack.dbias is not None, "sliced", pack.sliced, "n_use", pack.n_use, "n_bwd", pack.n_bwd)
        print("args", job["args"], "I16", ops.I16, "U8", ops.U8, "edge", job["idx"][3] is not None)
    return real()
ops.take_bias_bwd_job = spy
model.training_step(batch, 0).backward()
