"""Seeded synthetic check-in data in the reference's raw formats (S-FSQ / S-GOW / S-BIG).

There is no network and the Foursquare / poi_data archives are absent from the reference
(`README.md:40-43`, `.MISSING_LARGE_BLOBS`), so every test, fixture and bench run draws its inputs
from here.  Formats follow the reference's producers:

* a *trajectory* is the dict written by `gen_pickles.py:820-832` (`node_name`, `edge_type`, `target`,
  `time`, `time_normal`, `user`, `cat`), which `owndata.py:343-357` turns into a PyG-like item
  (`x [N,1]`, `edge_index [2,E]`, `edge_attr [E]`, `y [1]`, `time [N,1]`, `time_normal [N,1]`,
  `user [1,1]`, `cat [N,1]`);
* a *POI universe* stands in for `Graph_{poi,adj,dist,cat}.csv` (`foursquare_process.py:648-754`)
  and `poi_data/<city>_distance.pkl` (a `(P+1)x(P+1)` matrix whose row/col 0 is the pad POI).

Only numpy is used so that the generator is bit-reproducible on any host.
"""
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import List, Optional

import numpy as np


def haversine_km(lat1, lon1, lat2, lon2):
    r = 6371.0
    p1, p2 = np.radians(lat1), np.radians(lat2)
    dp = p2 - p1
    dl = np.radians(lon2) - np.radians(lon1)
    a = np.sin(dp / 2) ** 2 + np.cos(p1) * np.cos(p2) * np.sin(dl / 2) ** 2
    return 2 * r * np.arcsin(np.sqrt(np.clip(a, 0.0, 1.0)))


@dataclass
class Universe:
    """Everything `model_fqandtoyo.Graphormer.__init__` reads from disk, as arrays."""
    P: int
    n_cat: int
    n_user: int
    poi_table: np.ndarray          # [P, 6] float64: POI ID, checkin_cnt, lat, lon, cat, check_freq  (Graph_poi.csv)
    graph_adj: np.ndarray          # [P, P] float32 transition counts                               (Graph_adj.csv)
    graph_dist: np.ndarray         # [P, P] float32 0/1 "within 3 km"                               (Graph_dist.csv)
    graph_cat: np.ndarray          # [n_cat, n_cat] float32 category transition counts              (Graph_cat.csv)
    distance: Optional[np.ndarray] = None   # [(P+1), (P+1)] float64 km, row/col 0 = pad           (<city>_distance.pkl)
    poi_columns: tuple = ("POI ID", "checkin_cnt", "lat", "lon", "cat", "check_freq")

    @property
    def cat_of_poi(self):
        return self.poi_table[:, 4].astype(np.int64)


def make_universe(P=64, n_cat=8, n_user=8, seed=0, with_distance=True, adj_per_row=6) -> Universe:
    rng = np.random.RandomState(seed)
    lat = 35.68 + 0.08 * rng.randn(P)
    lon = 139.76 + 0.10 * rng.randn(P)
    cat = rng.randint(1, n_cat + 1, size=P)
    cat[:n_cat] = np.arange(1, n_cat + 1)          # every category occurs (OneHotEncoder width == n_cat)
    checkin_cnt = rng.randint(1, 200, size=P)
    check_freq = rng.randint(1, 10, size=P)
    poi_table = np.stack([np.arange(1, P + 1), checkin_cnt, lat, lon, cat, check_freq], 1).astype(np.float64)

    if with_distance:
        d = haversine_km(lat[:, None], lon[:, None], lat[None, :], lon[None, :])
        d = 0.5 * (d + d.T)
        np.fill_diagonal(d, 0.0)
        distance = np.zeros((P + 1, P + 1), dtype=np.float64)
        distance[1:, 1:] = d
        graph_dist = ((d <= 3.0) & (d > 0)).astype(np.float32)
    else:
        distance = None
        # banded stand-in with the same density scale, no O(P^2) trig
        graph_dist = np.zeros((P, P), dtype=np.float32)
        for k in range(1, min(P, 16)):
            idx = np.arange(P - k)
            graph_dist[idx, idx + k] = 1.0
            graph_dist[idx + k, idx] = 1.0

    graph_adj = np.zeros((P, P), dtype=np.float32)
    rows = np.repeat(np.arange(P), adj_per_row)
    cols = rng.randint(0, P, size=rows.size)
    np.add.at(graph_adj, (rows, cols), 1.0)
    graph_cat = rng.randint(0, 5, size=(n_cat, n_cat)).astype(np.float32)
    return Universe(P=P, n_cat=n_cat, n_user=n_user, poi_table=poi_table, graph_adj=graph_adj,
                    graph_dist=graph_dist, graph_cat=graph_cat, distance=distance)


@dataclass
class SparseUniverse:
    """A POI universe too large for dense P x P matrices (S-BIG, BASELINE configs[4]: P = 100 000).  Same role as
    `Universe`, with `graph_dist` as a scipy CSR 0/1 matrix ("within `radius_km`", foursquare_process.py:648-754) and the
    (P+1) x (P+1) distance pickle replaced by what it is computed from -- coordinates -- plus fixed bin edges."""
    P: int
    n_cat: int
    n_user: int
    poi_table: np.ndarray          # [P, 6] float64: POI ID, checkin_cnt, lat, lon, cat, check_freq
    graph_dist: object             # scipy.sparse.csr_matrix [P, P] float32 0/1
    graph_cat: np.ndarray          # [n_cat, n_cat] float32
    coords: np.ndarray             # [P+1, 2] float64 lat, lon; row 0 = pad POI
    bin_edges: np.ndarray          # [num_bins + 1] float64 km: poi_pos = np.digitize(haversine, bin_edges)
    num_bins: int
    distance: Optional[np.ndarray] = None
    graph_adj: Optional[np.ndarray] = None
    poi_columns: tuple = ("POI ID", "checkin_cnt", "lat", "lon", "cat", "check_freq")

    @property
    def cat_of_poi(self):
        return self.poi_table[:, 4].astype(np.int64)


def make_sparse_universe(P=100000, n_cat=300, n_user=1080, seed=0, radius_km=3.0, target_degree=32, num_bins=64) -> SparseUniverse:
    """Seeded synthetic city whose "within radius_km" graph has ~target_degree neighbours per POI whatever P is (the
    Gaussian spread grows with sqrt(P): avg neighbours = r^2 P / (4 sigma^2)); neighbours from a k-d tree, never P^2."""
    from scipy import sparse
    from scipy.spatial import cKDTree
    rng = np.random.RandomState(seed)
    sigma_km = radius_km * np.sqrt(P / (4.0 * target_degree))
    xy = rng.randn(P, 2) * sigma_km                                   # local east / north kilometres
    lat = 35.68 + xy[:, 1] / 110.574
    lon = 139.76 + xy[:, 0] / (111.320 * np.cos(np.radians(35.68)))
    cat = rng.randint(1, n_cat + 1, size=P)
    cat[:n_cat] = np.arange(1, n_cat + 1)
    checkin_cnt = rng.randint(1, 200, size=P)
    check_freq = rng.randint(1, 10, size=P)
    poi_table = np.stack([np.arange(1, P + 1), checkin_cnt, lat, lon, cat, check_freq], 1).astype(np.float64)
    pairs = cKDTree(xy).query_pairs(r=radius_km, output_type="ndarray")
    # keep exactly the pairs whose HAVERSINE distance is within the radius (what the collator's poi_pos is computed from)
    d = haversine_km(lat[pairs[:, 0]], lon[pairs[:, 0]], lat[pairs[:, 1]], lon[pairs[:, 1]])
    pairs = pairs[(d <= radius_km) & (d > 0)]
    rows = np.concatenate([pairs[:, 0], pairs[:, 1]])
    cols = np.concatenate([pairs[:, 1], pairs[:, 0]])
    graph_dist = sparse.csr_matrix((np.ones(rows.size, dtype=np.float32), (rows, cols)), shape=(P, P))
    graph_dist.sum_duplicates()
    graph_dist.data[:] = 1.0
    graph_cat = rng.randint(0, 5, size=(n_cat, n_cat)).astype(np.float32)
    coords = np.zeros((P + 1, 2), dtype=np.float64)
    coords[1:, 0], coords[1:, 1] = lat, lon
    # bin edges from a sample of pair distances (the reference derives them from the full matrix it cannot have here)
    a, b = rng.randint(0, P, size=200000), rng.randint(0, P, size=200000)
    sample = haversine_km(lat[a], lon[a], lat[b], lon[b])
    bin_edges = np.linspace(0.0, float(sample.max()) * 1.25, num_bins + 1)
    return SparseUniverse(P=P, n_cat=n_cat, n_user=n_user, poi_table=poi_table, graph_dist=graph_dist, graph_cat=graph_cat,
                          coords=coords, bin_edges=bin_edges, num_bins=num_bins)


def densify(uni: SparseUniverse) -> Universe:
    """The same universe with dense matrices (small P only): what the oracle / the dense product path consume."""
    lat, lon = uni.coords[1:, 0], uni.coords[1:, 1]
    d = haversine_km(lat[:, None], lon[:, None], lat[None, :], lon[None, :])
    distance = np.zeros((uni.P + 1, uni.P + 1), dtype=np.float64)
    distance[1:, 1:] = d
    return Universe(P=uni.P, n_cat=uni.n_cat, n_user=uni.n_user, poi_table=uni.poi_table,
                    graph_adj=np.zeros((uni.P, uni.P), dtype=np.float32), graph_dist=np.asarray(uni.graph_dist.todense(), dtype=np.float32),
                    graph_cat=uni.graph_cat, distance=distance)


def sample_num_nodes(rng, n, dist="fsq", lo=2, hi=256):
    """Per-trajectory node counts.  'fsq': clip(round(exp(N(1.5,1.2))), 2, 256)."""
    if dist == "fsq":
        v = np.rint(np.exp(rng.normal(1.5, 1.2, size=n))).astype(np.int64)
        return np.clip(v, lo, hi)
    raise ValueError(dist)


def make_trajectory(rng, P, n_nodes, n_user, cat_of_poi=None, extra_visits=0.3):
    """One raw trajectory dict (`gen_pickles.py:820-832`): distinct POIs, edge = transition count."""
    n_nodes = int(min(n_nodes, P))
    pois = rng.choice(P, size=n_nodes, replace=False) + 1          # node_name: 1..P, distinct
    # a walk that visits every node once in order, then revisits: chain-like digraph with long SPDs
    length = n_nodes + rng.poisson(extra_visits * n_nodes)
    walk = list(range(n_nodes))
    cur = n_nodes - 1
    for _ in range(length - n_nodes):
        # mostly local jumps so that the graph stays chain-like
        if rng.rand() < 0.7:
            nxt = int(np.clip(cur + rng.randint(-3, 4), 0, n_nodes - 1))
        else:
            nxt = int(rng.randint(0, n_nodes))
        walk.append(nxt)
        cur = nxt
    edge_type = np.zeros((n_nodes, n_nodes), dtype=np.int64)
    for a, b in zip(walk[:-1], walk[1:]):
        edge_type[a, b] += 1                                         # self transitions allowed (diag)
    edge_type = np.minimum(edge_type, 47)
    slot = rng.randint(0, 48, size=n_nodes)
    time_normal = np.where(slot == 0, 0.0, slot / 48.0).astype(np.float32)   # gen_pickles.py:805-809
    target = int(rng.randint(1, P + 1))
    user = int(rng.randint(0, n_user))
    cat = (cat_of_poi[pois - 1] if cat_of_poi is not None else rng.randint(1, 9, size=n_nodes)).astype(np.int64)
    return dict(node_name=pois.astype(np.int64), edge_type=edge_type, target=np.array([target], dtype=np.int64),
                time=slot.astype(np.int64), time_normal=time_normal, user=np.array([user], dtype=np.int64), cat=cat)


def trajectory_to_item(traj, idx=0):
    """Raw dict -> PyG-like item, exactly as `owndata.py:343-357` does (torch tensors)."""
    import torch
    x = torch.from_numpy(traj["node_name"]).to(torch.long).view(-1, 1)
    y = torch.from_numpy(traj["target"]).to(torch.long)
    adj = torch.from_numpy(traj["edge_type"])
    edge_index = adj.nonzero(as_tuple=False).t().contiguous()
    edge_attr = adj[edge_index[0], edge_index[1]].to(torch.long)
    item = SimpleNamespace(x=x, y=y, edge_index=edge_index, edge_attr=edge_attr)
    item.time = torch.from_numpy(traj["time"]).to(torch.long).view(-1, 1)
    item.time_normal = torch.from_numpy(traj["time_normal"]).to(torch.float).view(-1, 1)
    item.user = torch.from_numpy(traj["user"]).to(torch.long).view(-1, 1)
    item.cat = torch.from_numpy(traj["cat"]).to(torch.long).view(-1, 1)
    item.idx = idx
    return item


def make_batch_of_trajectories(seed, G, P, n_user, cat_of_poi=None, n_nodes: Optional[List[int]] = None,
                               dist="fsq", hi=256):
    rng = np.random.RandomState(seed)
    if n_nodes is None:
        n_nodes = sample_num_nodes(rng, G, dist=dist, hi=hi)
    return [make_trajectory(rng, P, int(n), n_user, cat_of_poi) for n in n_nodes]


def random_digraph(rng, n, p_edge=0.15, max_count=6):
    """Dense count matrix of a random digraph (for SPD property tests)."""
    m = (rng.rand(n, n) < p_edge).astype(np.int64) * rng.randint(1, max_count + 1, size=(n, n))
    return m
