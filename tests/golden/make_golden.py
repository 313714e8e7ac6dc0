"""Regenerate the committed golden vectors by RUNNING THE REFERENCE (only possible in the build
container, where /root/reference is mounted).  Usage:  python tests/golden/make_golden.py [g1 g2 ...]

Fixtures are data: seeded inputs plus the reference's outputs for them.  Model weights are not
stored; they are re-created by `fill_params` below from numpy's MT19937 stream (platform
independent), in `named_parameters()` order, so a fixture is a few hundred KB.

G1 algos.floyd_warshall / gen_edge_input          
G2 wrapper.preprocess_item + collator.collator    (rows 8-9)
G3 collator_foursquare / collator_gowalla          (row 10)
G4 EncoderLayer fwd+bwd, model.py and fq variants  (rows 1-3)
G5 assembled graph_attn_bias, both variants        (rows 4-5)
G6 end-to-end logits / loss / grads, both models   (rows 11-14)
G7 PolynomialDecayLR, GradientTailLoss, get_acc    (row 14 + §8f)
G8 real Gowalla trajectories + universe            (make_golden_real.py, its own script)
G9 EncoderLayer with a `mask`                      (rows 1-2, model.py:446-448)
"""
import os
import sys
import pickle
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import  # noqa: E402

WS = os.path.join(_ref_import.SCRATCH, "ws")


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.0f} KB, {len(arrays)} arrays")


# --------------------------------------------------------------------------------------------- G1
def crafted_graphs():
    g = {}
    # chain 0->1->2->3->4 with distinct counts
    a = np.zeros((5, 5), np.int64)
    for i in range(4):
        a[i, i + 1] = i + 1
    g["chain5"] = a
    # path through node 0 (truncation quirk): 1->0 (5), 0->2 (7)
    a = np.zeros((3, 3), np.int64)
    a[1, 0] = 5
    a[0, 2] = 7
    g["via_node0"] = a
    # same, relabelled 0->1->2
    a = np.zeros((3, 3), np.int64)
    a[0, 1] = 5
    a[1, 2] = 7
    g["chain3"] = a
    # two components + 2-cycle + self loop
    a = np.zeros((6, 6), np.int64)
    a[0, 1] = 2
    a[1, 0] = 3
    a[2, 2] = 4
    a[3, 4] = 1
    a[4, 5] = 1
    a[5, 3] = 9
    g["components"] = a
    g["single"] = np.zeros((1, 1), np.int64)
    g["single_loop"] = np.array([[3]], np.int64)
    g["two_nodes_noedge"] = np.zeros((2, 2), np.int64)
    # longer path through 0 in the middle: 3->2->0->1->4
    a = np.zeros((5, 5), np.int64)
    a[3, 2] = 1
    a[2, 0] = 2
    a[0, 1] = 3
    a[1, 4] = 4
    g["mid_node0"] = a
    return g


def ref_preprocess_arrays(algos, counts):
    """adjacency / edge features exactly as wrapper.preprocess_item builds them (wrapper.py:42-60)."""
    n = counts.shape[0]
    adj = counts != 0
    feat = np.zeros((n, n, 1), np.int64)
    feat[adj, 0] = counts[adj] + 2
    M, path = algos.floyd_warshall(adj)
    max_dist = int(np.amax(M))
    ei = algos.gen_edge_input(max_dist, path, feat)
    return adj, feat, M, path, max_dist, ei


def make_g1(algos):
    from mobgt_amd import synth
    out = {}
    names = []
    graphs = crafted_graphs()
    rng = np.random.RandomState(11)
    for n in (2, 5, 17, 40, 94):
        traj = synth.make_trajectory(rng, 200, n, 4)
        graphs[f"walk{n}"] = traj["edge_type"]
    for n, p in ((7, 0.3), (23, 0.12), (40, 0.06)):
        graphs[f"rand{n}"] = synth.random_digraph(rng, n, p)
    for name, counts in graphs.items():
        adj, feat, M, path, max_dist, ei = ref_preprocess_arrays(algos, counts)
        names.append(name)
        out[f"{name}/counts"] = counts.astype(np.int16)
        out[f"{name}/M"] = M.astype(np.int16)
        out[f"{name}/path"] = path.astype(np.int16)
        out[f"{name}/max_dist"] = np.array(max_dist)
        out[f"{name}/edge_input20"] = ei[:, :, :20, :].astype(np.int16)
        out[f"{name}/edge_input_shape"] = np.array(ei.shape)
        # everything past column 20 must be -1 or a continuation; keep a checksum of the full tensor
        out[f"{name}/edge_input_sum"] = np.array(ei.astype(np.float64).sum())
    # N > 510: a 600-node directed cycle (sentinel arithmetic case) -- M/path only
    n = 600
    counts = np.zeros((n, n), np.int64)
    counts[np.arange(n), (np.arange(n) + 1) % n] = 1
    M, path = algos.floyd_warshall(counts != 0)
    names.append("cycle600")
    out["cycle600/counts"] = counts.astype(np.int16)
    out["cycle600/M"] = M.astype(np.int16)
    out["cycle600/path"] = path.astype(np.int16)
    out["names"] = np.array(names)
    save("g1_algos.npz", **out)


# --------------------------------------------------------------------------------------- workspace
def write_universe(uni, city_pkls=("tky_distance.pkl", "gowalla_distance.pkl", "toyota_distance.pkl")):
    """Materialise a synthetic POI universe where the reference looks for it (relative to cwd)."""
    import pandas as pd
    os.makedirs(os.path.join(WS, "graphormer"), exist_ok=True)
    os.makedirs(os.path.join(WS, "dataset", "poi_data"), exist_ok=True)
    for ds in ("foursquaregraph", "gowalla_nevda", "toyotagraph"):
        raw = os.path.join(WS, "dataset", ds, "raw")
        os.makedirs(raw, exist_ok=True)
        df = pd.DataFrame(uni.poi_table, columns=list(uni.poi_columns))
        for c in ("POI ID", "checkin_cnt", "cat", "check_freq"):
            df[c] = df[c].astype(np.int64)
        df.to_csv(os.path.join(raw, "Graph_poi.csv"), index=False)
        pd.DataFrame(uni.graph_adj).to_csv(os.path.join(raw, "Graph_adj.csv"), index=False)
        pd.DataFrame(uni.graph_dist).to_csv(os.path.join(raw, "Graph_dist.csv"), index=False)
        pd.DataFrame(uni.graph_cat).to_csv(os.path.join(raw, "Graph_cat.csv"), index=False)
    for f in city_pkls:
        with open(os.path.join(WS, "dataset", "poi_data", f), "wb") as fh:
            pickle.dump(uni.distance, fh)


@contextlib.contextmanager
def in_ws():
    old = os.getcwd()
    os.chdir(os.path.join(WS, "graphormer"))
    try:
        yield
    finally:
        os.chdir(old)


def traj_arrays(prefix, trajs):
    out = {}
    for i, t in enumerate(trajs):
        for k, v in t.items():
            out[f"{prefix}{i}/{k}"] = np.asarray(v)
    out[f"{prefix}count"] = np.array(len(trajs))
    return out


def batch_arrays(prefix, b, skip=()):
    out = {}
    for k, v in vars(b).items():
        if k in skip:
            continue
        a = v.numpy()
        if a.dtype == np.int64:
            a = a.astype(np.int32)
        out[f"{prefix}{k}"] = a
    return out


# --------------------------------------------------------------------------------------------- G2/G3
def make_g2_g3():
    import copy
    import torch
    from mobgt_amd import synth
    import wrapper
    import collator as rcoll

    uni = synth.make_universe(P=64, n_cat=8, n_user=8, seed=3)
    write_universe(uni)
    trajs = synth.make_batch_of_trajectories(seed=5, G=8, P=64, n_user=8, cat_of_poi=uni.cat_of_poi,
                                             n_nodes=[3, 9, 2, 14, 6, 1, 11, 5])
    out = traj_arrays("traj", trajs)
    items = [wrapper.preprocess_item(synth.trajectory_to_item(t, idx=i)) for i, t in enumerate(trajs)]
    for i, it in enumerate(items):
        out[f"item{i}/rel_pos"] = it.rel_pos.numpy().astype(np.int16)
        out[f"item{i}/edge_input_shape"] = np.array(it.edge_input.shape)
        out[f"item{i}/edge_input20"] = it.edge_input[:, :, :20, :].numpy().astype(np.int16)
        out[f"item{i}/in_degree"] = it.in_degree.numpy().astype(np.int16)
        out[f"item{i}/out_degree"] = it.out_degree.numpy().astype(np.int16)
        out[f"item{i}/x"] = it.x.numpy().astype(np.int32)
        out[f"item{i}/user"] = it.user.numpy().astype(np.int32)
        out[f"item{i}/attn_edge_type"] = it.attn_edge_type.numpy().astype(np.int16)
        out[f"item{i}/adj"] = it.adj.numpy()
        out[f"item{i}/adj1"] = it.adj1.numpy()
        out[f"item{i}/attn_bias"] = it.attn_bias.numpy()
    b = rcoll.collator(copy.deepcopy(items), max_node=512, multi_hop_max_dist=20, rel_pos_max=1024)
    out.update(batch_arrays("stock/", b))
    # rel_pos_max masking + max_node filtering active
    b2 = rcoll.collator(copy.deepcopy(items), max_node=12, multi_hop_max_dist=5, rel_pos_max=3)
    out.update(batch_arrays("stock_masked/", b2))
    save("g2_collator.npz", **out)

    out3 = traj_arrays("traj", trajs)
    out3["distance"] = uni.distance.astype(np.float64)
    with in_ws():
        bf = rcoll.collator_foursquare(copy.deepcopy(items), max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)
        bg = rcoll.collator_gowalla(copy.deepcopy(items), max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)
    d = uni.distance
    dm = np.delete(np.delete(d, 0, axis=0), 0, axis=1)
    nb, edges = rcoll.freedman_diaconis_bins(dm - dm.min(), True)
    out3["num_bins"] = np.array(nb)
    out3["bin_edges"] = edges
    # feature_matrix (Laplacian eigenvectors, complex, never read by the model) is kept only as shape
    out3["fsq/feature_matrix_shape"] = np.array(bf.feature_matrix.shape)
    out3.update(batch_arrays("fsq/", bf, skip=("feature_matrix",)))
    out3.update(batch_arrays("gow/", bg, skip=("feature_matrix",)))
    save("g3_collator_fq.npz", **out3)


if __name__ == "__main__":
    which = set(sys.argv[1:]) or {"g1", "g2", "g4", "g5", "g6", "g7", "g9", "g11"}
    algos = _ref_import.install()
    if "g1" in which:
        make_g1(algos)
    if "g2" in which or "g3" in which:
        make_g2_g3()
    if {"g4", "g5", "g6", "g7", "g9", "g11"} & which:
        import make_golden_model
        make_golden_model.run(which)
