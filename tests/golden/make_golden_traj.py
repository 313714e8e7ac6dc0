"""G10: a TRAINING TRAJECTORY of the reference on real Gowalla data ("accuracy parity of a trained model,
not just forward parity").

The reference's fq Graphormer (gowalla_nevda, hidden 128, 6 layers, 8 heads, ffn 1024 = BASELINE configs[2]) is trained by the
reference's own pieces -- `training_step` (model_fqandtoyo.py:1434-1478), `configure_optimizers` (:1599-1616: AdamW +
PolynomialDecayLR stepped every update, lr.py:17-31) -- for 30 updates on real trajectories of `raw/train.pickle`, batches of
16 through `wrapper.preprocess_item` + `collator.collator_gowalla`, every dropout off (the module in eval() mode: the
constructor-constant GCN / positional dropouts cannot be set from the arguments).  Then it is evaluated on 256 real
trajectories of `raw/test.pickle` with the reference's `get_acc` / `MRR_metric` and `test_epoch_end`'s bookkeeping (:1546-1597).

Stored in `tests/golden/g10_traj.npz`: the 480 + 256 raw trajectories (those with at most 64 nodes, first in sorted (user,
key) order), the loss and learning rate of every update, the values of a fixed sample of every parameter after update 30, the
test logits of the first 16 test trajectories, and the metrics.  The POI universe is golden G8's (same archive, same haversine
stand-in for the missing distance pickle): the tests take it from `g8_gowalla_real.npz`.

    python tests/golden/make_golden_traj.py
"""
import copy
import io
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import  # noqa: E402
from make_golden import save, in_ws, traj_arrays  # noqa: E402
from make_golden_real import as_raw, ref_item, setup_real_workspace  # noqa: E402

STEPS, BATCH, N_TEST, MAX_N, SEED = 30, 16, 256, 64, 83
TRAJ_ARGS = dict(n_layers=6, num_heads=8, hidden_dim=128, dropout_rate=0.0, intput_dropout_rate=0.0, weight_decay=0.01,
                 ffn_dim=1024, dataset_name="gowalla_nevda", warmup_updates=10, tot_updates=100, peak_lr=1e-3, end_lr=1e-9,
                 edge_type="multi_hop", multi_hop_max_dist=20, attention_dropout_rate=0.0)


def param_sample(p):
    """What the fixture keeps of a parameter: all of it up to 512 elements, else 512 evenly spaced elements of the flat tensor."""
    f = np.asarray(p).reshape(-1)
    return f if f.size <= 512 else f[np.linspace(0, f.size - 1, 512).astype(np.int64)]


def first_keys(data, n, max_n):
    keys = []
    for u in sorted(data):
        for k in sorted(data[u]):
            if int(data[u][k]["node_name"].numel()) <= max_n:
                keys.append((u, k))
                if len(keys) == n:
                    return keys
    raise RuntimeError("not enough trajectories")


def main():
    import torch
    _ref_import.install()
    import wrapper
    import collator as rcoll
    import model_fqandtoyo as rfq
    from inputs import fill_params
    from make_golden_model import cpu_cuda_alias

    files, poi_df, poi, gdist, gcat, dist = setup_real_workspace()
    train = pickle.load(io.BytesIO(files["gowalla_nevda/raw/train.pickle"]))
    test = pickle.load(io.BytesIO(files["gowalla_nevda/raw/test.pickle"]))
    ktr, kte = first_keys(train, STEPS * BATCH, MAX_N), first_keys(test, N_TEST, MAX_N)
    mtr, mte = [train[u][k] for u, k in ktr], [test[u][k] for u, k in kte]
    out = {}
    out.update(traj_arrays("train/traj", [as_raw(m) for m in mtr]))
    out.update(traj_arrays("test/traj", [as_raw(m) for m in mte]))
    out["args/steps_batch_ntest"] = np.array([STEPS, BATCH, N_TEST])
    out["args/lr"] = np.array([TRAJ_ARGS["warmup_updates"], TRAJ_ARGS["tot_updates"], TRAJ_ARGS["peak_lr"], TRAJ_ARGS["end_lr"],
                               TRAJ_ARGS["weight_decay"]])
    out["seed"] = np.array(SEED)

    base_init = torch.optim.lr_scheduler.LRScheduler.__init__
    # torch >= 2.7 dropped the `verbose` positional the reference still passes (lr.py:15)
    torch.optim.lr_scheduler.LRScheduler.__init__ = lambda self, o, last_epoch=-1, verbose=False: base_init(self, o, last_epoch)
    with in_ws():
        m = rfq.Graphormer(**TRAJ_ARGS).eval()
    fill_params(m, SEED)
    out["param_names"] = np.array([n for n, _ in m.named_parameters()])
    out["param_shapes"] = np.array([str(tuple(p.shape)) for _, p in m.named_parameters()])
    (opt,), (sch,) = m.configure_optimizers()
    sched = sch["scheduler"]
    losses, lrs = [], []
    for s in range(STEPS):
        items = [wrapper.preprocess_item(ref_item(mol, s * BATCH + i)) for i, mol in enumerate(mtr[s * BATCH:(s + 1) * BATCH])]
        with in_ws():
            b = rcoll.collator_gowalla(copy.deepcopy(items), max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)
        assert int(b.poi_pos.max()) < m.poi_pos_encoder.weight.shape[0]
        lrs.append(opt.param_groups[0]["lr"])
        with cpu_cuda_alias():
            loss = m.training_step(b, s)
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        losses.append(loss.item())
        print("step", s, "lr %.3e" % lrs[-1], "loss %.6f" % losses[-1], "padded N", b.x.shape[1], flush=True)
    out["losses"] = np.array(losses, dtype=np.float64)
    out["lrs"] = np.array(lrs, dtype=np.float64)
    for pn, p in m.named_parameters():
        out[f"final/{pn}"] = param_sample(p.detach().numpy())
    # ---- evaluation on real test trajectories: test_step + test_epoch_end's bookkeeping with the reference's metric functions
    tot, mrr, n = np.zeros(8), 0.0, 0
    with torch.no_grad():
        for s in range(N_TEST // BATCH):
            items = [wrapper.preprocess_item(ref_item(mol, s * BATCH + i)) for i, mol in enumerate(mte[s * BATCH:(s + 1) * BATCH])]
            with in_ws():
                b = rcoll.collator_gowalla(copy.deepcopy(items), max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)
            o = m.test_step(b, s)
            y_pred, y_true = o["y_pred"][0], o["y_true"]
            if s == 0:
                out["test/logits0"] = y_pred.numpy()
            a, d = rfq.get_acc(y_true, y_pred)
            a, d = np.asarray(a, dtype=np.float64).reshape(4), np.asarray(d, dtype=np.float64).reshape(4)
            tot += np.array([a[2], a[1], a[0], d[2], d[1], d[0], a[3], d[3]])
            mrr += float(rfq.MRR_metric(y_true, y_pred))
            n += len(y_true)
    out["metrics/acc1_5_10_ndcg1_5_10_acc20_ndcg20"] = tot / n
    out["metrics/mrr"] = np.array(mrr / n)
    print("metrics", tot / n, "mrr", mrr / n)
    save("g10_traj.npz", **out)


if __name__ == "__main__":
    main()
