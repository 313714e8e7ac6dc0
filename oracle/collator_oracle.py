"""ORACLE (test infrastructure): restatement of the reference's per-sample preprocessing and batch
collation -- `wrapper.py:18-102` and `collator.py:11-458` -- in plain torch/numpy on the CPU.

Pinned against tests/golden/g2_collator.npz and g3_collator_fq.npz (outputs of the reference).
The product never imports this module.
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import algos_oracle as algos


def convert_to_single_emb(x, offset=512):
    """wrapper.py:18-22"""
    feature_num = x.size(1) if len(x.size()) > 1 else 1
    feature_offset = 1 + torch.arange(0, feature_num * offset, offset, dtype=torch.long)
    return x + feature_offset


def preprocess_item(item):
    """wrapper.py:25-102: item(x, edge_index, edge_attr, user, ...) -> item + SPD / edge-path fields."""
    edge_attr, edge_index, x = item.edge_attr, item.edge_index, item.x
    if edge_attr is None:
        edge_attr = torch.zeros((edge_index.shape[1]), dtype=torch.long)            # :31-33
    N = x.size(0)
    x = convert_to_single_emb(x)                                                     # :37
    user = convert_to_single_emb(item.user)                                          # :39
    adj_orig = torch.zeros([N, N], dtype=torch.bool)                                 # :42-43
    adj_orig[edge_index[0, :], edge_index[1, :]] = True
    if len(edge_attr.size()) == 1:
        edge_attr = edge_attr[:, None]
    attn_edge_type = torch.zeros([N, N, edge_attr.size(-1)], dtype=torch.long)      # :49-53
    attn_edge_type[edge_index[0, :], edge_index[1, :]] = convert_to_single_emb(edge_attr) + 1
    shortest_path_result, path = algos.floyd_warshall(adj_orig.numpy())              # :55
    max_dist = np.amax(shortest_path_result)                                         # :58 (510 if any unreachable)
    edge_input = algos.gen_edge_input(max_dist, path, attn_edge_type.numpy())        # :60
    rel_pos = torch.from_numpy(shortest_path_result).long()                          # :61
    attn_bias = torch.zeros([N + 1, N + 1], dtype=torch.float)                       # :63-65
    adj = torch.zeros([N + 1, N + 1], dtype=torch.bool)                              # :67-70
    adj[edge_index[0, :], edge_index[1, :]] = True
    adj1 = torch.zeros([N, N], dtype=torch.bool)                                     # :73-76
    adj1[edge_index[0, :], edge_index[1, :]] = True
    adj[N, :] = True                                                                 # :79-81
    adj[:, N] = True
    out = SimpleNamespace(**vars(item))
    out.user = user
    out.adj1 = adj1
    out.x = x
    out.attn_bias = attn_bias
    out.attn_edge_type = attn_edge_type
    out.rel_pos = rel_pos
    out.in_degree = adj_orig.long().sum(dim=1).view(-1)                              # :97 (row sums, named "in")
    out.out_degree = adj_orig.long().sum(dim=0).view(-1)                             # :98
    out.edge_input = torch.from_numpy(edge_input).long()                             # :99
    out.adj = adj
    return out


# ---- collator.py:11-101 padding helpers -------------------------------------------------------
def pad_1d_unsqueeze(x, padlen):
    x = x + 1
    new_x = x.new_zeros([padlen], dtype=x.dtype)
    new_x[: x.size(0)] = x
    return new_x.unsqueeze(0)


def pad_2d_squeeze(x, padlen):
    x = x - 1
    new_x = x.new_zeros([padlen, x.size(1)], dtype=x.dtype)
    new_x[: x.size(0), :] = x
    return new_x.unsqueeze(0)


def pad_time_unsqueeze(x, padlen):
    new_x = x.new_zeros([padlen, x.size(1)], dtype=x.dtype)
    new_x[: x.size(0), :] = x
    return new_x.unsqueeze(0)


def pad_2d_bool(x, padlen):
    new_x = x.new_zeros([padlen, padlen], dtype=x.dtype)
    n = x.size(0)
    new_x[:n, :n] = x
    return new_x.unsqueeze(0)


def pad_attn_bias_unsqueeze(x, padlen):
    n = x.size(0)
    if n < padlen:
        new_x = x.new_zeros([padlen, padlen], dtype=x.dtype).fill_(float("-inf"))
        new_x[:n, :n] = x
        new_x[n:, :n] = 0                                                            # collator.py:62
        x = new_x
    return x.unsqueeze(0)


def pad_edge_type_unsqueeze(x, padlen):
    n = x.size(0)
    new_x = x.new_zeros([padlen, padlen, x.size(-1)], dtype=x.dtype)
    new_x[:n, :n, :] = x
    return new_x.unsqueeze(0)


def pad_rel_pos_unsqueeze(x, padlen):
    x = x + 1
    n = x.size(0)
    new_x = x.new_zeros([padlen, padlen], dtype=x.dtype)
    new_x[:n, :n] = x
    return new_x.unsqueeze(0)


def pad_3d_unsqueeze(x, padlen1, padlen2, padlen3):
    x = x + 1
    l1, l2, l3, l4 = x.size()
    new_x = x.new_zeros([max(l1, padlen1), max(l2, padlen2), max(l3, padlen3), l4], dtype=x.dtype)
    new_x[:l1, :l2, :l3, :] = x
    return new_x.unsqueeze(0)


def _apply_rel_pos_max(attn_biases, rel_poses, rel_pos_max):
    for b, r in zip(attn_biases, rel_poses):                                         # collator.py:247-251
        b[1:, 1:][r >= rel_pos_max] = float("-inf")


def collator(items, max_node=512, multi_hop_max_dist=20, rel_pos_max=20):
    """collator.py:218-299 (stock Graphormer batch).  Returns a namespace with Batch's attributes."""
    items = [it for it in items if it is not None and it.x.size(0) <= max_node]
    attn_biases = [it.attn_bias.clone() for it in items]
    rel_poses = [it.rel_pos for it in items]
    edge_inputs = [it.edge_input[:, :, :multi_hop_max_dist, :] for it in items]
    _apply_rel_pos_max(attn_biases, rel_poses, rel_pos_max)
    max_node_num = max(it.x.size(0) for it in items)
    max_node_num = 4 * (max_node_num // 4) + 3                                       # :259-260
    max_dist = max(e.size(-2) for e in edge_inputs)
    return SimpleNamespace(
        idx=torch.LongTensor([it.idx for it in items]),
        y=torch.cat([it.y + 1 for it in items]),                                     # :264
        x=torch.cat([pad_2d_squeeze(it.x, max_node_num) for it in items]),
        edge_input=torch.cat([pad_3d_unsqueeze(e, max_node_num, max_node_num, max_dist) for e in edge_inputs]),
        attn_bias=torch.cat([pad_attn_bias_unsqueeze(b, max_node_num + 1) for b in attn_biases]),
        adj=torch.cat([pad_2d_bool(it.adj, max_node_num + 1) for it in items]),
        attn_edge_type=torch.cat([pad_edge_type_unsqueeze(it.attn_edge_type, max_node_num + 1) for it in items]),
        rel_pos=torch.cat([pad_rel_pos_unsqueeze(r, max_node_num) for r in rel_poses]),
        in_degree=torch.cat([pad_1d_unsqueeze(it.in_degree, max_node_num) for it in items]),
        out_degree=torch.cat([pad_1d_unsqueeze(it.out_degree, max_node_num) for it in items]),
    )


def freedman_diaconis_bins(x, return_bins=False):
    """collator.py:301-308 (len(x) = number of ROWS of the matrix)."""
    iqr = np.subtract(*np.percentile(x, [75, 25]))
    binsize = 2 * iqr * np.power(len(x), -1 / 3)
    bins = np.ceil((np.max(x) - np.min(x)) / binsize)
    if return_bins:
        return int(bins), np.histogram(x, int(bins))[1]
    return int(bins)


def collator_poi(items, poi_distance_matrix, max_node=512, multi_hop_max_dist=20, rel_pos_max=20):
    """collator.py:310-458 / 460-608 (collator_foursquare == collator_gowalla up to the pickle name).
    `poi_distance_matrix` is the unpickled (P+1)x(P+1) matrix the reference re-loads every batch
    (:429 / :579).  feature_matrix (Laplacian eigenvectors, :393-410) is never read by the model and
    is not restated."""
    items = [it for it in items if it is not None and it.x.size(0) <= max_node]
    attn_biases = [it.attn_bias.clone() for it in items]
    rel_poses = [it.rel_pos for it in items]
    edge_inputs = [it.edge_input[:, :, :multi_hop_max_dist, :] for it in items]
    _apply_rel_pos_max(attn_biases, rel_poses, rel_pos_max)
    max_node_num = max(it.x.size(0) for it in items)                                 # no 4k+3 rounding (:362-364)
    max_dist = max(e.size(-2) for e in edge_inputs)
    x = torch.cat([pad_2d_squeeze(it.x, max_node_num) for it in items])
    rel_pos = torch.cat([pad_rel_pos_unsqueeze(r, max_node_num) for r in rel_poses])
    indx = (x != 0).sum(dim=-2)                                                      # :424-426
    poi_pos = torch.cat([pad_rel_pos_unsqueeze(r, max_node_num) for r in rel_poses])  # :428
    dm = np.delete(np.delete(poi_distance_matrix, 0, axis=0), 0, axis=1)             # :430-432
    _, bins = freedman_diaconis_bins(dm - dm.min(), True)
    for i in range(x.size(0)):                                                       # :435-437
        n = int(indx[i])
        for j in range(n):
            row = [poi_distance_matrix[int(x[i][j])][int(x[i][k])] for k in range(n)]
            poi_pos[i][j][:n] = torch.LongTensor(np.digitize(row, bins))
    return SimpleNamespace(
        idx=torch.LongTensor([it.idx for it in items]),
        y=torch.cat([it.y for it in items]),                                         # :367 (unshifted)
        x=x,
        time=torch.cat([pad_time_unsqueeze(it.time, max_node_num) for it in items]),
        time_normal=torch.cat([pad_time_unsqueeze(it.time_normal, max_node_num) for it in items]),
        user=torch.cat([it.user for it in items]),
        cat=torch.cat([pad_time_unsqueeze(it.cat, max_node_num) for it in items]),
        edge_input=torch.cat([pad_3d_unsqueeze(e, max_node_num, max_node_num, max_dist) for e in edge_inputs]),
        attn_bias=torch.cat([pad_attn_bias_unsqueeze(b, max_node_num + 1) for b in attn_biases]),
        adj=torch.cat([pad_2d_bool(it.adj, max_node_num + 1) for it in items]),
        adj1=torch.cat([pad_2d_bool(it.adj1, max_node_num) for it in items]),
        attn_edge_type=torch.cat([pad_edge_type_unsqueeze(it.attn_edge_type, max_node_num + 1) for it in items]),
        rel_pos=rel_pos,
        in_degree=torch.cat([pad_1d_unsqueeze(it.in_degree, max_node_num) for it in items]),
        out_degree=torch.cat([pad_1d_unsqueeze(it.out_degree, max_node_num) for it in items]),
        poi_pos=poi_pos,
        bins=bins,
    )
