"""Debug probe: what of a fresh batch costs GPU time in EpochLoop -- variants of the loop with parts switched off (the skipped
parts leave stale data behind: timing only)."""
import os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from mobgt_amd import workloads
from mobgt_amd.train import EpochLoop
uni, model, coll = workloads.build("fsq", "cuda", seed=1)
pool = workloads.make_pool("fsq", 64, 16, uni, seed0=5000)
data = [t for trajs in pool for t in trajs]
loop = EpochLoop(model, coll, data, batch_size=16, seed=1)
loop.run_epoch(0); loop.run_epoch(1); loop.run_epoch(2)
torch.cuda.synchronize()
pc = time.perf_counter

def run(tag, eps=(20, 21, 22, 23, 24)):
    torch.cuda.synchronize()
    t0 = pc(); n = 0
    for ep in eps:
        n += loop.run_epoch(ep)["steps"]
    torch.cuda.synchronize()
    print("%-46s %.4f ms/step (%d steps)" % (tag, (pc() - t0) / n * 1e3, n), flush=True)

run("fresh batch (as shipped)")
run("fresh batch (again)")
orig_finish = coll.finish_into
coll.finish_into = lambda views, work: work
run("no collate kernels on the copy stream")
coll.finish_into = orig_finish

orig_launch = loop._launch
def launch_nocopy(self, slot, st):
    cur = torch.cuda.current_stream()
    cur.wait_event(st["ready"])
    if st["free"] is None:
        st["free"] = torch.cuda.Event()
    st["free"].record(cur)
    return self.ts.step(slot["index"])
loop._launch = types.MethodType(launch_nocopy, loop)
run("no device-to-device copy between replays")
def launch_nowait(self, slot, st):
    cur = torch.cuda.current_stream()
    n = slot["copy_bytes"]
    slot["buf"][:n].copy_(st["dev"][:n], non_blocking=True)
    if st["free"] is None:
        st["free"] = torch.cuda.Event()
    st["free"].record(cur)
    return self.ts.step(slot["index"])
loop._launch = types.MethodType(launch_nowait, loop)
run("no wait on the copy stream's event")
def launch_only(self, slot, st):
    return self.ts.step(slot["index"])
loop._launch = types.MethodType(launch_only, loop)
run("replay only (copy stream still works)")
coll.finish_into = lambda views, work: work
run("replay only, no collate kernels")
orig_stage = loop._stage
def stage_nocopy(self, ids):
    from mobgt_amd.data import bucket_nodes
    trajs = [self.dataset[i] for i in ids]
    G = len(trajs)
    N = bucket_nodes(max(len(t["node_name"]) for t in trajs), self.buckets)
    slot = self._slot(G, N)
    return slot, slot["stages"][0]
loop._stage = types.MethodType(stage_nocopy, loop)
run("replay only, no host work at all")
