"""The fork-safe host half of the boundary (include/mobgt_cpu.h -> mobgt_amd/libmobgt_cpu.so): symbols, bit-exactness
against the reference's own outputs (golden G1 / G2) and against the C oracle on random digraphs, the reference's error
behaviour, and -- the reason it exists -- use from forked DataLoader workers exactly as the reference wires it
(`wrapper.py:55-60` under `data.py:282-295`: `DataLoader(num_workers=..., collate_fn=partial(collator_x, ...))`)."""
import os
import re
from functools import partial

import numpy as np
import pytest
import torch

from mobgt_amd import _lib_cpu, algos, synth, wrapper
from mobgt_amd import collator as pc
from oracle import algos_oracle as ao

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_library_exports_every_declared_symbol_and_has_no_hip_dependency():
    _lib_cpu.build()
    hdr = open(os.path.join(ROOT, "include", "mobgt_cpu.h")).read()
    declared = set(re.findall(r"\b(mobgt_[a-z0-9_]+)\s*\(", hdr))
    handle = _lib_cpu.lib()
    for name in declared:
        assert hasattr(handle, name), name
    assert declared == set(_lib_cpu.SIGNATURES)
    assert handle.mobgt_cpu_abi_version() == 1
    import subprocess
    needed = subprocess.run(["readelf", "-d", _lib_cpu.LIB_PATH], capture_output=True, text=True).stdout
    assert "amdhip" not in needed and "hsa" not in needed and "gomp" not in needed, needed


def test_backend_is_host_without_an_initialised_gpu(monkeypatch):
    monkeypatch.delenv("MOBGT_ALGOS_BACKEND", raising=False)
    if not torch.cuda.is_initialized():
        assert algos.backend() == "host"
    monkeypatch.setenv("MOBGT_ALGOS_BACKEND", "device")
    assert algos.backend() == "device"


def test_host_algos_match_reference_g1(golden_dir, monkeypatch):
    monkeypatch.setenv("MOBGT_ALGOS_BACKEND", "host")
    z = np.load(os.path.join(golden_dir, "g1_algos.npz"))
    for name in z["names"]:
        c = z[f"{name}/counts"].astype(np.int64)
        M, p = algos.floyd_warshall(c != 0)
        assert M.dtype == np.int64 and p.dtype == np.int64
        assert np.array_equal(M, z[f"{name}/M"]) and np.array_equal(p, z[f"{name}/path"]), name
        if name == "cycle600":
            continue
        n = c.shape[0]
        feat = np.zeros((n, n, 1), np.int64)
        feat[c != 0, 0] = c[c != 0] + 2
        ei = algos.gen_edge_input(int(M.max()), p, feat)
        assert ei.dtype == np.float32 and tuple(ei.shape) == tuple(z[f"{name}/edge_input_shape"])
        assert np.array_equal(ei[:, :, :20], z[f"{name}/edge_input20"]), name
        assert ei.astype(np.float64).sum() == float(z[f"{name}/edge_input_sum"])


def test_host_algos_match_c_oracle_on_random_digraphs_and_keep_error_behaviour(monkeypatch):
    monkeypatch.setenv("MOBGT_ALGOS_BACKEND", "host")
    rng = np.random.RandomState(11)
    for n, p_edge in ((1, 0.5), (2, 0.5), (9, 0.3), (33, 0.1), (64, 0.05), (130, 0.02), (300, 0.008)):
        c = synth.random_digraph(rng, n, p_edge)
        M, p = algos.floyd_warshall(c != 0)
        M2, p2 = ao.floyd_warshall(c != 0)
        assert np.array_equal(M, M2) and np.array_equal(p, p2), n
        feat = rng.randint(0, 50, size=(n, n, 2)).astype(np.int64)                  # F = 2: the general layout
        md = int(M.max())
        assert np.array_equal(algos.gen_edge_input(md, p, feat), ao.gen_edge_input(md, p, feat)), n
        for _ in range(10):
            i, j = rng.randint(0, n, 2)
            if i != j and p[i, j] != 510:
                assert algos.get_all_edges(p, i, j) == ao.get_all_edges(p, i, j)
    # integer weights other than 0/1 (the reference takes any int matrix): same arithmetic
    w = rng.randint(0, 4, size=(40, 40)).astype(np.int64)
    M, p = algos.floyd_warshall(w)
    M2, p2 = ao.floyd_warshall(w)
    assert np.array_equal(M, M2) and np.array_equal(p, p2)
    # a path longer than max_dist: IndexError, as the Cython bounds check raises (algos.pyx:94)
    chain = np.zeros((5, 5), np.int64)
    for i in range(4):
        chain[i, i + 1] = 1
    M, p = algos.floyd_warshall(chain)
    with pytest.raises(IndexError):
        algos.gen_edge_input(2, p, np.zeros((5, 5, 1), np.int64))
    # a path matrix that does not terminate: the reference recurses until RecursionError
    loop = np.zeros((3, 3), np.int64)
    loop[0, 2], loop[0, 1], loop[1, 2] = 1, 1, 1
    loop[0, 1] = 2                                   # 0->1 via 2, 0->2 via 1: expands forever
    with pytest.raises(RecursionError):
        algos.get_all_edges(loop, 0, 2)
    with pytest.raises(AssertionError):
        algos.floyd_warshall(np.zeros((2, 3), np.int64))


class _TrajDataset(torch.utils.data.Dataset):
    """What `MyFoursquareGraphDataset.__getitem__` does (wrapper.py:158-162): item -> preprocess_item(item)."""

    def __init__(self, trajs):
        self.trajs = trajs

    def __len__(self):
        return len(self.trajs)

    def __getitem__(self, i):
        return wrapper.preprocess_item(synth.trajectory_to_item(self.trajs[i], idx=i))


def test_preprocess_item_runs_in_forked_dataloader_workers(golden_dir, monkeypatch):
    """data.py:282-295 with num_workers = 2: preprocess_item (-> algos, host back end) + collator_foursquare inside
    forked workers give the reference's batch (golden G3) -- and no worker initialises the GPU."""
    monkeypatch.delenv("MOBGT_ALGOS_BACKEND", raising=False)
    z = np.load(os.path.join(golden_dir, "g3_collator_fq.npz"))
    trajs = [{k: z[f"traj{i}/{k}"] for k in ("node_name", "edge_type", "target", "time", "time_normal", "user", "cat")}
             for i in range(int(z["trajcount"]))]
    pc.register_poi_distance("tky_distance.pkl", z["distance"])
    n = len(trajs)
    dl = torch.utils.data.DataLoader(_TrajDataset(trajs), batch_size=n, shuffle=False, num_workers=2,
                                     multiprocessing_context="fork",
                                     collate_fn=partial(pc.collator_foursquare, max_node=30000, multi_hop_max_dist=20,
                                                        rel_pos_max=1024))
    batches = list(dl)
    assert len(batches) == 1
    b = batches[0]
    fields = ("attn_bias", "attn_edge_type", "rel_pos", "in_degree", "out_degree", "x", "edge_input", "y", "adj", "time",
              "adj1", "time_normal", "user", "cat", "poi_pos")
    for f in fields:
        ref, got = z[f"fsq/{f}"], getattr(b, f).numpy()
        assert got.shape == ref.shape, f
        assert np.array_equal(got.astype(ref.dtype) if ref.dtype.kind != "f" else got, ref), f
