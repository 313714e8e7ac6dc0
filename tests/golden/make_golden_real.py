"""G8: golden vectors from REAL Gowalla trajectories ("real Gowalla graphs N in {1,2,5,17,94}").

Runs the reference on the data the reference ships -- `/root/reference/gowalla_nevda.7z`: raw/train.pickle (user ->
trajectory -> dict of tensors, gen_pickles.py:820-832) and the real POI universe raw/Graph_{poi,adj,dist,cat}.csv
(P = 3 679, 253 categories) -- and writes `tests/golden/g8_gowalla_real.npz`:

  batch A  eight real trajectories, N = 1, 2, 5, 17, 94 (the sizes SURVEY names) + 8, 12, 30; items built exactly as
           owndata.py:343-357 does; the reference's `algos.floyd_warshall` / `gen_edge_input`, `wrapper.preprocess_item`,
           `collator.collator_gowalla`, then `model_fqandtoyo.Graphormer` (gowalla_nevda, hidden 128, 6 layers, 8 heads,
           ffn 1024, multi_hop_max_dist 20 = BASELINE configs[2]) forward + training_step + backward;
  batch B  the 329-node trajectory (the shortest of the seven real graphs with N >= 300) and a 5-node one: collated
           indices and eval logits.

The one file the archive does NOT hold is `poi_data/gowalla_distance.pkl` (README.md:40-43).  Stand-in: the haversine
distance between the POIs' real coordinates, rounded to metres (`real_distance` below; the rounding makes the matrix
bit-reproducible on any host, so the tests rebuild it from the fixture's coordinates instead of storing 108 MB).  The real
`Graph_dist.csv` (symmetric 0/1, "within 3 km") is stored as the packed bits of its upper triangle.

    python tests/golden/make_golden_real.py
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
from make_golden import save, in_ws, batch_arrays, traj_arrays, WS  # noqa: E402
from sevenz_min import read_archive  # noqa: E402

ARCHIVE = "/root/reference/gowalla_nevda.7z"
SIZES_A = (1, 2, 5, 17, 94, 8, 12, 30)
REAL_ARGS = dict(n_layers=6, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1,
                 weight_decay=0.01, ffn_dim=1024, dataset_name="gowalla_nevda", warmup_updates=10, tot_updates=100,
                 peak_lr=2e-4, end_lr=1e-9, edge_type="multi_hop", multi_hop_max_dist=20, attention_dropout_rate=0.1)
SEED = 81


def real_distance(poi_table):
    """(P+1) x (P+1) float64 km, row / column 0 = the pad POI: haversine of the real coordinates rounded to 1 m."""
    from mobgt_amd.synth import haversine_km
    lat, lon = poi_table[:, 2], poi_table[:, 3]
    d = np.round(haversine_km(lat[:, None], lon[:, None], lat[None, :], lon[None, :]), 3)
    out = np.zeros((len(lat) + 1, len(lat) + 1), dtype=np.float64)
    out[1:, 1:] = d
    return out


def pick(data, sizes):
    """First trajectory (users, then trajectory keys, in sorted order) of each requested node count."""
    by_n = {}
    for u in sorted(data):
        for k in sorted(data[u]):
            by_n.setdefault(int(data[u][k]["node_name"].numel()), []).append((u, k))
    return [by_n[n][0] for n in sizes], by_n


def as_raw(t):
    """The pickle's tensors as the numpy dict `mobgt_amd.synth.trajectory_to_item` / DeviceCollator take."""
    return dict(node_name=t["node_name"].numpy().astype(np.int64), edge_type=t["edge_type"].numpy().astype(np.int64),
                target=t["target"].numpy().astype(np.int64), time=t["time"].numpy().astype(np.int64),
                time_normal=t["time_normal"].numpy().astype(np.float32), user=t["user"].numpy().astype(np.int64),
                cat=t["cat"].numpy().astype(np.int64))


def ref_item(mol, idx):
    """owndata.py:343-357 (GowallaGraph.process): the PyG `Data` fields the wrapper reads, on a plain namespace."""
    import torch
    from types import SimpleNamespace
    x = mol["node_name"].to(torch.long).view(-1, 1)
    y = mol["target"].to(torch.long)
    adj = mol["edge_type"]
    edge_index = adj.nonzero(as_tuple=False).t().contiguous()
    edge_attr = adj[edge_index[0], edge_index[1]].to(torch.long)
    data = SimpleNamespace(x=x, edge_index=edge_index, edge_attr=edge_attr, y=y)
    data.time = mol["time"].to(torch.long).view(-1, 1)
    data.time_normal = mol["time_normal"].to(torch.float).view(-1, 1)
    data.user = mol["user"].to(torch.long).view(-1, 1)
    data.cat = mol["cat"].to(torch.long).view(-1, 1)
    data.idx = idx
    return data


def setup_real_workspace():
    """The shipped archive's real POI universe written where the reference looks for it (relative to the workspace's
    graphormer/ directory) + the haversine stand-in for the one missing file; -> (archive files, poi frame, poi table, Graph_dist,
    Graph_cat, distance matrix)."""
    import pandas as pd
    files = read_archive(ARCHIVE)
    raw = os.path.join(WS, "dataset", "gowalla_nevda", "raw")
    os.makedirs(raw, exist_ok=True)
    os.makedirs(os.path.join(WS, "graphormer"), exist_ok=True)
    os.makedirs(os.path.join(WS, "dataset", "poi_data"), exist_ok=True)
    for n in ("Graph_poi", "Graph_adj", "Graph_dist", "Graph_cat"):
        with open(os.path.join(raw, n + ".csv"), "wb") as f:
            f.write(files[f"gowalla_nevda/raw/{n}.csv"])
    poi_df = pd.read_csv(os.path.join(raw, "Graph_poi.csv"))
    poi = poi_df.to_numpy().astype(np.float64)
    gdist = pd.read_csv(os.path.join(raw, "Graph_dist.csv")).to_numpy()
    gcat = pd.read_csv(os.path.join(raw, "Graph_cat.csv")).to_numpy()
    dist = real_distance(poi)
    d = dist[1:, 1:]
    print("Graph_dist.csv vs (0 < d <= 3 km) of the rounded stand-in:", int(((gdist != 0) != ((d <= 3.0) & (d > 0))).sum()), "pairs differ")
    assert np.array_equal(gdist, gdist.T) and set(np.unique(gdist)) <= {0.0, 1.0}
    with open(os.path.join(WS, "dataset", "poi_data", "gowalla_distance.pkl"), "wb") as f:
        pickle.dump(dist, f)
    return files, poi_df, poi, gdist, gcat, dist


def main():
    import pandas as pd
    import torch
    algos = _ref_import.install()
    import wrapper
    import collator as rcoll
    import model_fqandtoyo as rfq
    from inputs import fill_params
    from make_golden_model import cpu_cuda_alias, grad_sample

    files, poi_df, poi, gdist, gcat, dist = setup_real_workspace()

    data = pickle.load(io.BytesIO(files["gowalla_nevda/raw/train.pickle"]))
    keys_a, by_n = pick(data, SIZES_A)
    big_n = min(n for n in by_n if n >= 300)
    keys_b = [by_n[big_n][0], by_n[5][1]]

    out = {}
    out["uni/poi_table"] = poi
    out["uni/poi_columns"] = np.array(list(poi_df.columns))
    out["uni/graph_cat"] = gcat.astype(np.float32)
    iu = np.triu_indices(gdist.shape[0], 1)
    out["uni/graph_dist_triu_bits"] = np.packbits((gdist[iu] != 0).astype(np.uint8))
    out["keys_a"] = np.array(keys_a)
    out["keys_b"] = np.array(keys_b)

    # ---------------------------------------------------------------- per-item: algos + preprocess_item (G1 / G2 on real graphs)
    mols_a = [data[u][k] for u, k in keys_a]
    mols_b = [data[u][k] for u, k in keys_b]
    out.update(traj_arrays("a/traj", [as_raw(m) for m in mols_a]))
    out.update(traj_arrays("b/traj", [as_raw(m) for m in mols_b]))
    items_a = [wrapper.preprocess_item(ref_item(m, i)) for i, m in enumerate(mols_a)]
    items_b = [wrapper.preprocess_item(ref_item(m, i)) for i, m in enumerate(mols_b)]
    for tag, items, mols in (("a", items_a, mols_a), ("b", items_b, mols_b)):
        for i, (it, mol) in enumerate(zip(items, mols)):
            adj = mol["edge_type"].numpy() != 0
            M, path = algos.floyd_warshall(adj)
            p = f"{tag}/item{i}/"
            out[p + "M"] = M.astype(np.int16)
            out[p + "path"] = path.astype(np.int16)
            out[p + "rel_pos"] = it.rel_pos.numpy().astype(np.int16)
            out[p + "edge_input_shape"] = np.array(it.edge_input.shape)
            ei = it.edge_input[:, :, :20, :].numpy().astype(np.int16)
            out[p + "edge_input20"] = ei if ei.shape[0] <= 100 else ei[::7]          # every 7th row of the 329-node graph
            out[p + "edge_input_sum"] = np.array(it.edge_input.numpy().astype(np.float64).sum())
            out[p + "in_degree"] = it.in_degree.numpy().astype(np.int16)
            out[p + "out_degree"] = it.out_degree.numpy().astype(np.int16)
            out[p + "x"] = it.x.numpy().astype(np.int32)
            out[p + "user"] = it.user.numpy().astype(np.int32)

    # ---------------------------------------------------------------- collator_gowalla + fq Graphormer on the real universe
    with in_ws():
        ba = rcoll.collator_gowalla(copy.deepcopy(items_a), max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)
        bb = rcoll.collator_gowalla(copy.deepcopy(items_b), max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)
        m = rfq.Graphormer(**REAL_ARGS).eval()
    dm = np.delete(np.delete(dist, 0, axis=0), 0, axis=1)
    nb, edges = rcoll.freedman_diaconis_bins(dm - dm.min(), True)
    out["num_bins"] = np.array(nb)
    out["bin_edges"] = edges
    assert m.poi_pos_encoder.weight.shape[0] == nb
    for tag, b in (("a", ba), ("b", bb)):
        assert int(b.poi_pos.max()) < nb, "poi_pos reaches the reference's latent out-of-bounds bin (SURVEY App. A)"
        arrs = batch_arrays(f"{tag}/batch/", b, skip=("feature_matrix", "adj", "adj1", "attn_edge_type"))
        if tag == "b":                                   # [2, 331, 331, 20, 1]: keep every 7th query row + the checksum
            ei = arrs.pop("b/batch/edge_input")
            arrs["b/batch/edge_input_rows7"] = ei[:, ::7].astype(np.int16)
            arrs["b/batch/edge_input_sum"] = np.array(ei.astype(np.float64).sum())
        out.update(arrs)
    fill_params(m, SEED)
    out["seed"] = np.array(SEED)
    out["param_names"] = np.array([n for n, _ in m.named_parameters()])
    out["param_shapes"] = np.array([str(tuple(p.shape)) for _, p in m.named_parameters()])
    captured = {}
    h = m.layers[0].register_forward_pre_hook(lambda mod, a: captured.__setitem__("bias", a[1].detach().clone()))
    with torch.no_grad():
        oa = m(ba)
        bias_a = captured["bias"].numpy()
        ob = m(bb)
        bias_b = captured["bias"].numpy()
    h.remove()
    out["a/bias_rows7"] = bias_a[:, :, ::7]
    out["b/bias_rows31_h0"] = bias_b[:, :1, ::31]
    out["a/logits"], out["a/cat_logits"] = oa[0].numpy(), oa[1].numpy()
    out["b/logits"], out["b/cat_logits"] = ob[0].numpy(), ob[1].numpy()
    m.zero_grad()
    with cpu_cuda_alias():
        loss = m.training_step(ba, 0)
    loss.backward()
    out["a/loss"] = np.array(loss.item())
    for pn, p in m.named_parameters():
        if p.grad is None:
            out[f"a/grad_none/{pn}"] = np.array(1)
        else:
            g = p.grad.double()
            out[f"a/gstat/{pn}"] = np.array([g.sum().item(), g.norm().item()])
            if p.grad.numel() <= 40000:
                out[f"a/grad/{pn}"] = grad_sample(p.grad.numpy())
    # the edge tables' gradients pass through the reference's explicit .half() casts (model_fqandtoyo.py:1178-1198): at the plain
    # loss, per-pair gradients below fp16's 6e-8 flush to zero INSIDE THE REFERENCE.  The reference trains with --precision 16
    # (README.md:62), i.e. under Lightning's GradScaler, whose initial scale is 65536: the same backward at loss x 65536 is what
    # its training sees -- stored (divided by the scale again) for those two tables
    m.zero_grad()
    with cpu_cuda_alias():
        loss_s = m.training_step(ba, 0) * 65536.0
    loss_s.backward()
    for pn, p in m.named_parameters():
        if pn.startswith("edge_") and p.grad is not None:
            g = p.grad.double() / 65536.0
            out[f"a_s65536/gstat/{pn}"] = np.array([g.sum().item(), g.norm().item()])
            out[f"a_s65536/grad/{pn}"] = grad_sample((p.grad / 65536.0).numpy())
    save("g8_gowalla_real.npz", **out)
    print("batch A nodes", [int(t["node_name"].numel()) for t in mols_a], "padded", tuple(ba.x.shape),
          "| batch B", [int(t["node_name"].numel()) for t in mols_b], tuple(bb.x.shape), "| bins", nb,
          "| loss", float(loss), "| max |logit|", float(oa[0].abs().max()))


if __name__ == "__main__":
    main()
