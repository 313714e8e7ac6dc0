"""Fixture generator: node-count histogram of the reference's Gowalla trajectory graphs (SURVEY §8d "S-GOW").

Reads `/root/reference/gowalla_nevda.7z` (raw/train.pickle + raw/test.pickle: user -> trajectory -> dict of tensors,
`gen_pickles.py:820-832`) in the build container and writes `tests/golden/gowalla_n_hist.npz`:
    n_values, n_counts   -- distinct node counts N over train + test graphs and how many graphs have each
    P, n_cat, n_user     -- universe sizes (rows of Graph_poi.csv, distinct categories, users)
    max_edge_count       -- largest transition count on an edge
The fixture is data (counts), ~1 KB.  `mobgt_amd.workloads` draws S-GOW batch shapes from it.

    python tests/golden/make_gowalla_hist.py
"""
import io
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from sevenz_min import read_archive  # noqa: E402


def main():
    files = read_archive("/root/reference/gowalla_nevda.7z")
    ns, max_edge = [], 0
    users = set()
    for part in ("train", "test"):
        data = pickle.load(io.BytesIO(files[f"gowalla_nevda/raw/{part}.pickle"]))
        for u, trajs in data.items():
            users.add(int(u))
            for t in trajs.values():
                ns.append(int(t["node_name"].numel()))
                max_edge = max(max_edge, int(t["edge_type"].max()))
    poi = np.genfromtxt(io.BytesIO(files["gowalla_nevda/raw/Graph_poi.csv"]), delimiter=",", skip_header=1)
    vals, counts = np.unique(np.array(ns), return_counts=True)
    out = os.path.join(HERE, "gowalla_n_hist.npz")
    np.savez_compressed(out, n_values=vals.astype(np.int32), n_counts=counts.astype(np.int32), P=np.int64(poi.shape[0]),
                        n_cat=np.int64(len(np.unique(poi[:, 4]))), n_user=np.int64(len(users)),
                        max_edge_count=np.int64(max_edge))
    n = np.array(ns)
    print(f"{len(ns)} graphs, N mean {n.mean():.2f} p50 {np.percentile(n, 50):.0f} p90 {np.percentile(n, 90):.0f} "
          f"p99 {np.percentile(n, 99):.0f} max {n.max()}; P {poi.shape[0]}, users {len(users)}, max edge count {max_edge}")
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
