import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from mobgt_amd import ops, workloads
uni, model, coll = workloads.build("fsq", "cuda", seed=1, model_overrides=dict(n_layers=2))
batch = coll(workloads.make_pool("fsq", 1, 16, uni)[0])
model.train()
res = {}
for tag, off in (("fused", "0"), ("split", "1"), ("split2", "1"), ("fused2", "0")):
    os.environ["MOBGT_NO_TOKEN_FWD_CHAIN"] = off
    ops.set_dropout_state(torch.tensor([3], dtype=torch.int64, device="cuda"), 11)
    with torch.no_grad():
        ops.front_deferral(False)
        idx = model.gather_indices(batch)
        out = model.node_features(batch, indices=idx)
        logits = model(batch)[0]
    torch.cuda.synchronize()
    q = getattr(out, "_mobgt_qkv", None)
    res[tag] = (out.float().clone(), (q if q is not None else out).float().clone(), logits.float().clone())
for a, b in (("fused", "split"), ("split", "split2"), ("fused", "fused2")):
    print(a, b, [float((x - y).abs().max()) for x, y in zip(res[a], res[b])])
