"""Drop-in for `graphormer/wrapper.py:18-102` (`convert_to_single_emb`, `preprocess_item`).

`preprocess_item(item)` keeps the reference's per-item contract -- same attributes, shapes, dtypes and
index shifts -- with the shortest-path work done by the HIP kernels behind `mobgt_amd.algos`.  The
PyG/ogb dataset subclasses of the reference (wrapper.py:105-196) are storage plumbing and out of scope;
the batched device pipeline that the trainer and bench use is `mobgt_amd.data.DeviceCollator`.
"""
import numpy as np
import torch

from . import algos


def convert_to_single_emb(x, offset=512):
    """wrapper.py:18-22"""
    feature_num = x.size(1) if len(x.size()) > 1 else 1
    feature_offset = 1 + torch.arange(0, feature_num * offset, offset, dtype=torch.long)
    return x + feature_offset


def preprocess_item(item):
    """wrapper.py:25-102"""
    edge_attr, edge_index, x = item.edge_attr, item.edge_index, item.x
    if edge_attr is None:
        edge_attr = torch.zeros((edge_index.shape[1]), dtype=torch.long)
    n = x.size(0)
    rows, cols = edge_index[0, :], edge_index[1, :]
    if len(edge_attr.size()) == 1:
        edge_attr = edge_attr[:, None]

    adj_orig = torch.zeros([n, n], dtype=torch.bool)
    adj_orig[rows, cols] = True
    attn_edge_type = torch.zeros([n, n, edge_attr.size(-1)], dtype=torch.long)
    attn_edge_type[rows, cols] = convert_to_single_emb(edge_attr) + 1           # :49-53

    shortest_path_result, path = algos.floyd_warshall(adj_orig.numpy())         # :55  (HIP)
    max_dist = np.amax(shortest_path_result)                                    # :58  (510 if any pair is unreachable)
    edge_input = algos.gen_edge_input(max_dist, path, attn_edge_type.numpy())   # :60  (HIP)

    adj = torch.zeros([n + 1, n + 1], dtype=torch.bool)                         # :67-81 (virtual token row/col)
    adj[rows, cols] = True
    adj[n, :] = True
    adj[:, n] = True

    item.x = convert_to_single_emb(x)                                           # :37
    item.user = convert_to_single_emb(item.user)                                # :39
    item.adj1 = adj_orig.clone()                                                # :73-76
    item.attn_bias = torch.zeros([n + 1, n + 1], dtype=torch.float)             # :63-65
    item.attn_edge_type = attn_edge_type
    item.rel_pos = torch.from_numpy(shortest_path_result).long()                # :61
    item.in_degree = adj_orig.long().sum(dim=1).view(-1)                        # :97 (row sums)
    item.out_degree = adj_orig.long().sum(dim=0).view(-1)                       # :98 (column sums)
    item.edge_input = torch.from_numpy(edge_input).long()                       # :99
    item.adj = adj
    return item
