"""Pins the oracle's data path (algos, preprocess_item, collators) to the reference's outputs
(tests/golden/g1..g3, produced by tests/golden/make_golden.py).  Integer work: bit-exact."""
import os

import numpy as np
import pytest
import torch

from mobgt_amd import synth
from oracle import algos_oracle as ao
from oracle import collator_oracle as co


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_algos_against_reference_g1(golden_dir):
    z = load(golden_dir, "g1_algos.npz")
    for name in z["names"]:
        c = z[f"{name}/counts"].astype(np.int64)
        M, p = ao.floyd_warshall(c != 0)
        assert np.array_equal(M, z[f"{name}/M"]), name
        assert np.array_equal(p, z[f"{name}/path"]), name
        if name == "cycle600":
            continue
        n = c.shape[0]
        feat = np.zeros((n, n, 1), np.int64)
        feat[c != 0, 0] = c[c != 0] + 2
        md = int(M.max())
        assert md == int(z[f"{name}/max_dist"])
        ei = ao.gen_edge_input(md, p, feat)
        assert tuple(ei.shape) == tuple(z[f"{name}/edge_input_shape"])
        assert np.array_equal(ei[:, :, :20], z[f"{name}/edge_input20"]), name
        assert ei.astype(np.float64).sum() == float(z[f"{name}/edge_input_sum"])


def test_known_answer_node0_truncation():
    """an intermediate node 0 is indistinguishable from 'no intermediate'."""
    c = np.zeros((3, 3), np.int64)
    c[1, 0], c[0, 2] = 5, 7
    M, p = ao.floyd_warshall(c != 0)
    assert M[1, 2] == 2 and p[1, 2] == 0
    feat = np.zeros((3, 3, 1), np.int64)
    feat[c != 0, 0] = c[c != 0] + 2
    ei = ao.gen_edge_input(int(M.max()), p, feat)
    assert ei[1, 2, :3, 0].tolist() == [0, -1, -1]
    assert ao.get_all_edges(p, 1, 2) == []


def _trajs(z):
    out = []
    for i in range(int(z["trajcount"])):
        out.append({k: z[f"traj{i}/{k}"] for k in ("node_name", "edge_type", "target", "time", "time_normal", "user", "cat")})
    return out


def _items(z):
    return [co.preprocess_item(synth.trajectory_to_item(t, idx=i)) for i, t in enumerate(_trajs(z))]


def _cmp_batch(z, prefix, b, fields):
    for f in fields:
        ref = z[f"{prefix}{f}"]
        got = getattr(b, f).numpy()
        assert got.shape == ref.shape, (f, got.shape, ref.shape)
        if ref.dtype.kind == "f":
            assert np.array_equal(got, ref), f          # only 0 / -inf / slot/48 values: exact
        else:
            assert np.array_equal(got.astype(np.int64), ref.astype(np.int64)), f


def test_preprocess_and_stock_collator_g2(golden_dir):
    z = load(golden_dir, "g2_collator.npz")
    items = _items(z)
    for i, it in enumerate(items):
        assert np.array_equal(it.rel_pos.numpy(), z[f"item{i}/rel_pos"])
        assert tuple(it.edge_input.shape) == tuple(z[f"item{i}/edge_input_shape"])
        assert np.array_equal(it.edge_input[:, :, :20].numpy(), z[f"item{i}/edge_input20"])
        for f in ("in_degree", "out_degree", "x", "user", "attn_edge_type", "adj", "adj1", "attn_bias"):
            assert np.array_equal(getattr(it, f).numpy(), z[f"item{i}/{f}"]), f
    fields = ("idx", "attn_bias", "attn_edge_type", "rel_pos", "in_degree", "out_degree", "x", "edge_input", "y", "adj")
    b = co.collator(items, max_node=512, multi_hop_max_dist=20, rel_pos_max=1024)
    _cmp_batch(z, "stock/", b, fields)
    b2 = co.collator(items, max_node=12, multi_hop_max_dist=5, rel_pos_max=3)
    _cmp_batch(z, "stock_masked/", b2, fields)


def test_poi_collators_g3(golden_dir):
    z = load(golden_dir, "g3_collator_fq.npz")
    items = _items(z)
    b = co.collator_poi(items, z["distance"], max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)
    assert np.array_equal(b.bins, z["bin_edges"])
    fields = ("idx", "attn_bias", "attn_edge_type", "rel_pos", "in_degree", "out_degree", "x", "edge_input", "y",
              "adj", "time", "adj1", "time_normal", "user", "cat", "poi_pos")
    _cmp_batch(z, "fsq/", b, fields)
    _cmp_batch(z, "gow/", b, fields)
