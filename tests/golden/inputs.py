"""Seeded input builders shared by the fixture generator (make_golden*.py, runs the reference) and
the parity tests (run the oracle / the HIP path).  numpy's MT19937 stream is platform independent,
so fixtures store only the reference's OUTPUTS plus the seed; inputs and weights are re-created."""
import numpy as np

ENCODER_CASES = [("c128_t5", 128, 5, 3, 1024), ("c128_t33", 128, 33, 2, 1024), ("c192_t33", 192, 33, 2, 1024),
                 ("c256_t130", 256, 130, 1, 1024), ("c192_t7", 192, 7, 4, 1024)]


def fill_params(module, seed, scale=0.08):
    """Deterministic weights: MT19937 normals in named_parameters() order; norm gains centred on 1."""
    import torch
    rng = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            v = rng.standard_normal(size=tuple(p.shape)).astype(np.float32) * scale
            if name.endswith("norm.weight") or name.endswith("ln.weight") or "norm1.weight" in name \
                    or "norm2.weight" in name:
                v = 1.0 + v
            p.copy_(torch.from_numpy(v))


def rand_bias(rng, G, H, T, n_real):
    """[G,H,T,T] bias with the collator's padding pattern: pad key columns are -inf for every row."""
    b = (rng.standard_normal((G, H, T, T)) * 0.7).astype(np.float32)
    for g in range(G):
        b[g, :, :, n_real[g]:] = -np.inf
    return b


def encoder_case(variant, C, T, G):
    """-> seed, x [G,T,C], bias [G,8,T,T], gy [G,T,C], n_real"""
    seed = (1000 * C + 10 * T + (1 if variant == "fq" else 0)) % 100003
    rng = np.random.RandomState(seed)
    x = rng.standard_normal((G, T, C)).astype(np.float32)
    n_real = [T] + [max(2, T - 1 - 3 * g) for g in range(1, G)]
    bias = rand_bias(rng, G, 8, T, n_real)
    gy = rng.standard_normal((G, T, C)).astype(np.float32)
    return seed, x, bias, gy, n_real


MASK_CASES = [("c128_t5", 128, 5, 3, 1024), ("c192_t33", 192, 33, 2, 1024)]


def mask_case(variant, C, T, G):
    """-> encoder_case(...) + mask [G,T,T] bool (about a fifth of the pairs; model.py:446-448)"""
    seed, x, bias, gy, n_real = encoder_case(variant, C, T, G)
    mask = np.random.RandomState(seed + 7).rand(G, T, T) < 0.2
    return seed, x, bias, gy, n_real, mask


# ------------------------------------------------------------------ G8: the real Gowalla universe (make_golden_real.py)
def real_distance(poi_table):
    """The stand-in for `poi_data/gowalla_distance.pkl` the G8 fixture was generated with: (P+1) x (P+1) float64 km,
    row / column 0 = the pad POI, haversine of the POIs' real coordinates rounded to 1 m (the rounding makes the matrix
    bit-reproducible across hosts; same expression as make_golden_real.real_distance)."""
    from mobgt_amd.synth import haversine_km
    lat, lon = poi_table[:, 2], poi_table[:, 3]
    d = np.round(haversine_km(lat[:, None], lon[:, None], lat[None, :], lon[None, :]), 3)
    out = np.zeros((len(lat) + 1, len(lat) + 1), dtype=np.float64)
    out[1:, 1:] = d
    return out


def real_universe(z):
    """g8_gowalla_real.npz -> synth.Universe of the real Gowalla POIs (P 3 679, 253 categories)."""
    from mobgt_amd.synth import Universe
    poi = z["uni/poi_table"]
    P = poi.shape[0]
    bits = np.unpackbits(z["uni/graph_dist_triu_bits"])[: P * (P - 1) // 2]
    gd = np.zeros((P, P), dtype=np.float32)
    gd[np.triu_indices(P, 1)] = bits
    gd = gd + gd.T
    return Universe(P=P, n_cat=int(len(np.unique(poi[:, 4]))), n_user=1080, poi_table=poi,
                    graph_adj=np.zeros((1, 1), dtype=np.float32), graph_dist=gd, graph_cat=z["uni/graph_cat"],
                    distance=real_distance(poi), poi_columns=tuple(str(c) for c in z["uni/poi_columns"]))


def real_trajs(z, tag):
    return [{k: z[f"{tag}/traj{i}/{k}"] for k in ("node_name", "edge_type", "target", "time", "time_normal", "user", "cat")}
            for i in range(int(z[f"{tag}/trajcount"]))]
