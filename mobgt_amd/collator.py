"""Drop-in for `graphormer/collator.py`: `Batch` / `Batch1` and the `collator*` functions with the same
signatures, field names, shapes, dtypes and index conventions (Appendix A of SURVEY.md).

The four near-identical collators of the reference (collator.py:218-299, 310-458, 460-608, 610-748)
share one implementation here.  Differences in HOW:
* the POI distance matrix is loaded once per process and its Freedman-Diaconis bin table cached,
  instead of un-pickling ~100 MB and recomputing percentiles for EVERY batch (collator.py:429-433);
* `poi_pos` is one vectorised `np.digitize` per graph instead of a G*N Python loop (:435-437);
* `feature_matrix` (Laplacian eigenvectors via torch.linalg.eig per graph, :393-410) is never read by
  the model; it is computed lazily on first attribute access so the field still exists.
"""
import os
import pickle

import numpy as np
import torch

POI_DATA_DIR = os.path.join("..", "dataset", "poi_data")
_POI_CACHE = {}


def freedman_diaconis_bins(x, return_bins=False):
    """collator.py:301-308"""
    iqr = np.subtract(*np.percentile(x, [75, 25]))
    binsize = 2 * iqr * np.power(len(x), -1 / 3)
    bins = np.ceil((np.max(x) - np.min(x)) / binsize)
    if return_bins:
        return int(bins), np.histogram(x, int(bins))[1]
    return int(bins)


def register_poi_distance(name, matrix):
    """Provide the (P+1)x(P+1) distance matrix for `name` ('tky_distance.pkl', ...) without a pickle on disk."""
    _POI_CACHE[name] = _prepare_distance(np.asarray(matrix))


def _prepare_distance(d):
    dm = np.delete(np.delete(d, 0, axis=0), 0, axis=1)                 # collator.py:430-432
    num_bins, edges = freedman_diaconis_bins(dm - dm.min(), True)
    return {"matrix": d, "num_bins": num_bins, "edges": edges}


def poi_distance(name):
    if name not in _POI_CACHE:
        with open(os.path.join(POI_DATA_DIR, name), "rb") as f:
            _POI_CACHE[name] = _prepare_distance(np.asarray(pickle.load(f, encoding="iso-8859-1")))
    return _POI_CACHE[name]


class Batch:
    """collator.py:103-146"""
    _fields = ("idx", "attn_bias", "attn_edge_type", "rel_pos", "in_degree", "out_degree", "x", "edge_input", "y", "adj")

    def __init__(self, **kw):
        for f in self._fields:
            setattr(self, f, kw[f])

    def to(self, device):
        for f in self._fields:
            setattr(self, f, getattr(self, f).to(device))
        return self

    def __len__(self):
        return self.in_degree.size(0)


class Batch1(Batch):
    """collator.py:149-215"""
    _fields = Batch._fields + ("adj1", "time", "time_normal", "user", "cat", "poi_pos")

    def __init__(self, **kw):
        super().__init__(**kw)
        self._feature_matrix = kw.get("feature_matrix")

    @property
    def feature_matrix(self):
        if self._feature_matrix is None:                                 # collator.py:393-410, on demand
            a = self.adj1.long().cpu()
            lap = torch.diag_embed(a.sum(dim=-1)) - a
            self._feature_matrix = torch.stack([torch.linalg.eig(m.float())[1] for m in lap]).to(self.adj1.device)
        return self._feature_matrix

    def to(self, device):
        super().to(device)
        if self._feature_matrix is not None:
            self._feature_matrix = self._feature_matrix.to(device)
        return self


def _pad(t, shape, fill=0):
    out = t.new_full(shape, fill)
    out[tuple(slice(0, s) for s in t.shape)] = t
    return out


def _collate(items, max_node, multi_hop_max_dist, rel_pos_max, round_nodes, y_shift, poi_pickle):
    items = [it for it in items if it is not None and it.x.size(0) <= max_node]
    G = len(items)
    biases = []
    for it in items:                                                     # rel_pos_max mask (collator.py:247-251)
        b = it.attn_bias.clone()
        b[1:, 1:][it.rel_pos >= rel_pos_max] = float("-inf")
        biases.append(b)
    N = max(it.x.size(0) for it in items)
    if round_nodes:
        N = 4 * (N // 4) + 3                                             # stock collator only (:259-260)
    T = N + 1
    edges = [it.edge_input[:, :, :multi_hop_max_dist, :] for it in items]
    D = max(e.size(-2) for e in edges)
    F = edges[0].size(-1)

    attn_bias = torch.full((G, T, T), float("-inf"))
    for g, b in enumerate(biases):                                       # pad_attn_bias_unsqueeze (:57-64)
        n = b.size(0)
        if n < T:
            attn_bias[g, :n, :n] = b
            attn_bias[g, n:, :n] = 0
        else:
            attn_bias[g] = b
    out = dict(
        idx=torch.LongTensor([it.idx for it in items]),
        attn_bias=attn_bias,
        attn_edge_type=torch.stack([_pad(it.attn_edge_type, (T, T, it.attn_edge_type.size(-1))) for it in items]),
        rel_pos=torch.stack([_pad(it.rel_pos + 1, (N, N)) for it in items]),          # +1, pad 0 (:76-83)
        in_degree=torch.stack([_pad(it.in_degree + 1, (N,)) for it in items]),        # (:11-18)
        out_degree=torch.stack([_pad(it.out_degree + 1, (N,)) for it in items]),
        x=torch.stack([_pad(it.x - 1, (N, it.x.size(1))) for it in items]),           # undo wrapper's +1 (:29-37)
        edge_input=torch.stack([_pad(e + 1, (N, N, D, F)) for e in edges]),           # (:86-93)
        y=torch.cat([it.y + y_shift for it in items]),
        adj=torch.stack([_pad(it.adj, (T, T), False) for it in items]),
    )
    if poi_pickle is None:
        return Batch(**out)
    out.update(
        adj1=torch.stack([_pad(it.adj1, (N, N), False) for it in items]),
        time=torch.stack([_pad(it.time, (N, it.time.size(1))) for it in items]),
        time_normal=torch.stack([_pad(it.time_normal, (N, it.time_normal.size(1))) for it in items]),
        user=torch.cat([it.user for it in items]),
        cat=torch.stack([_pad(it.cat, (N, it.cat.size(1))) for it in items]),
    )
    pd = poi_distance(poi_pickle)
    poi_pos = out["rel_pos"].clone()                                     # initialised from rel_pos+1 (:428)
    xs = out["x"][:, :, 0].numpy()
    for g in range(G):
        n = int((xs[g] != 0).sum())
        ids = xs[g, :n]
        poi_pos[g, :n, :n] = torch.from_numpy(np.digitize(pd["matrix"][np.ix_(ids, ids)], pd["edges"]))
    out["poi_pos"] = poi_pos
    return Batch1(**out)


def collator(items, max_node=512, multi_hop_max_dist=20, rel_pos_max=20):
    """collator.py:218-299"""
    return _collate(items, max_node, multi_hop_max_dist, rel_pos_max, round_nodes=True, y_shift=1, poi_pickle=None)


def collator_foursquare(items, max_node=512, multi_hop_max_dist=20, rel_pos_max=20):
    """collator.py:310-458"""
    return _collate(items, max_node, multi_hop_max_dist, rel_pos_max, False, 0, "tky_distance.pkl")


def collator_gowalla(items, max_node=512, multi_hop_max_dist=20, rel_pos_max=20):
    """collator.py:460-608"""
    return _collate(items, max_node, multi_hop_max_dist, rel_pos_max, False, 0, "gowalla_distance.pkl")


def collator_toyota(items, max_node=512, multi_hop_max_dist=20, rel_pos_max=20):
    """collator.py:610-748"""
    return _collate(items, max_node, multi_hop_max_dist, rel_pos_max, False, 0, "toyota_distance.pkl")
