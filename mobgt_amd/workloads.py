"""The benchmark workloads of SURVEY.md §8d as code, shared by `bench.py` and the parity tests so that the configuration
that is timed is the configuration that is checked against the oracle.

    fsq  S-FSQ  (BASELINE configs[1]): foursquaregraph, P = 7856 POIs, 300 categories, 1080 users; per-graph N ~
         clip(round(exp(Normal(1.5, 1.2))), 2, 256); hidden 128 (C = 192, d = 24), 6 layers, 8 heads.
    gow  S-GOW  (BASELINE configs[2]): gowalla_nevda, P = 3679, 253 categories, 1080 users; N drawn from the empirical
         node-count histogram of the reference's Gowalla graphs (tests/golden/gowalla_n_hist.npz, 6869 graphs, max 814).
    big  S-BIG  (BASELINE configs[4]): 16 graphs x 784 nodes per GPU, P = 100 000, hidden 192 (C = 256, d = 32),
         12 layers; the POI graph is held as CSR (a dense 100k x 100k adjacency does not exist anywhere), distance bins
         come from coordinates on the fly.

All three use the reference's README hyper-parameters (`README.md:62-69`): dropout 0.1 everywhere, AdamW, weight decay
0.01, PolynomialDecayLR 40k / 400k, peak 2e-4, multi_hop_max_dist 20, 16 trajectories per GPU per step.
"""
import os

import numpy as np
import torch

from . import synth

_HERE = os.path.dirname(os.path.abspath(__file__))
GOWALLA_HIST = os.path.join(os.path.dirname(_HERE), "tests", "golden", "gowalla_n_hist.npz")

COMMON = dict(num_heads=8, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01, ffn_dim=1024, warmup_updates=40000,
              tot_updates=400000, peak_lr=2e-4, end_lr=1e-9, edge_type="multi_hop", multi_hop_max_dist=20,
              attention_dropout_rate=0.1)

WORKLOADS = {
    "fsq": dict(model=dict(COMMON, n_layers=6, hidden_dim=128, dataset_name="foursquaregraph"), P=7856, n_cat=300, n_user=1080,
                n_dist="fsq", hi=256, label="S-FSQ (BASELINE configs[1])"),
    "gow": dict(model=dict(COMMON, n_layers=6, hidden_dim=128, dataset_name="gowalla_nevda"), P=3679, n_cat=253, n_user=1080,
                n_dist="gowalla", hi=1024, label="S-GOW (BASELINE configs[2])"),
    "big": dict(model=dict(COMMON, n_layers=12, hidden_dim=192, dataset_name="foursquaregraph"), P=100000, n_cat=300,
                n_user=1080, n_dist="fixed784", hi=784, label="S-BIG (BASELINE configs[4], per-GPU slice)"),
}
# the name bench.py's default run uses
FSQ_MODEL_ARGS = WORKLOADS["fsq"]["model"]


def gowalla_node_counts(n, seed):
    """n per-graph node counts drawn from the empirical Gowalla histogram by stratified inverse-CDF sampling (one draw
    per 1/n quantile slice, then shuffled): a 128-graph pool reproduces the distribution's body AND reaches into its
    tail (p99 = 92, max 814) instead of depending on the luck of 128 iid draws."""
    z = np.load(GOWALLA_HIST)
    vals, counts = z["n_values"].astype(np.int64), z["n_counts"].astype(np.float64)
    cdf = np.cumsum(counts) / counts.sum()
    rng = np.random.RandomState(seed)
    u = (np.arange(n) + rng.rand(n)) / n
    out = vals[np.minimum(np.searchsorted(cdf, u, side="left"), len(vals) - 1)]
    rng.shuffle(out)
    return np.maximum(out, 1)


def node_counts(name, G, seed):
    w = WORKLOADS[name]
    if w["n_dist"] == "fsq":
        return synth.sample_num_nodes(np.random.RandomState(seed), G, dist="fsq", hi=w["hi"])
    if w["n_dist"] == "fixed784":
        return np.full(G, 784, dtype=np.int64)
    raise ValueError(name)


def make_pool(name, n_batches, G, uni, seed0=1000):
    """`n_batches` lists of G raw trajectories (gen_pickles.py:820-832 format) for workload `name`."""
    w = WORKLOADS[name]
    pool = []
    if w["n_dist"] == "gowalla":
        ns = gowalla_node_counts(n_batches * G, seed0).reshape(n_batches, G)
    for i in range(n_batches):
        if w["n_dist"] == "fsq":
            # (the S-FSQ generator seeds the node counts and the trajectories from one stream, as round 1 did)
            trajs = synth.make_batch_of_trajectories(seed=seed0 + i, G=G, P=w["P"], n_user=w["n_user"],
                                                     cat_of_poi=uni.cat_of_poi, hi=w["hi"])
        else:
            n_nodes = ns[i] if w["n_dist"] == "gowalla" else node_counts(name, G, seed0 + i)
            trajs = synth.make_batch_of_trajectories(seed=seed0 + i, G=G, P=w["P"], n_user=w["n_user"],
                                                     cat_of_poi=uni.cat_of_poi, n_nodes=[int(n) for n in n_nodes])
        pool.append(trajs)
    return pool


def build(name, device, seed=1, dtype="bf16", gemm_dtype="bf16", fused=True, P=None, model_overrides=None, variant="fq"):
    """(universe, model on `device`, DeviceCollator) for workload `name`.  dtype "bf16": attention operands / bias /
    GCN adjacency product in bf16; gemm_dtype "bf16": the encoder layers' GEMM-facing activations too.
    variant "stock": `graphormer/model.py`'s Graphormer (pre-LN EncoderLayer, C = hidden_dim = 128, d = 16 -- the layer
    BASELINE.json's north_star names) on the same trajectories: atom / degree embeddings, the same bias assembly without
    the distance-bin table, a (P + 1)-way head on the graph token.  The reference itself cannot run model.py on the POI
    datasets (data.py:74 leaves `num_class` out; entry.py:10 imports model_fqandtoyo): the variant exists to time and
    check the stock layer stack under MobGT's batch shapes."""
    from .data import DeviceCollator, make_bin_table
    from .model_fqandtoyo import Graphormer
    w = WORKLOADS[name]
    P = int(P or w["P"])
    bf16 = dtype == "bf16"
    torch.manual_seed(seed)
    args = dict(w["model"])
    args.update(model_overrides or {})
    if variant == "stock":
        from .model import Graphormer as StockGraphormer
        if name == "big":
            raise ValueError("variant 'stock' is defined for the fsq / gow workloads")
        uni = synth.make_universe(P=P, n_cat=w["n_cat"], n_user=w["n_user"], seed=seed)
        _, _, table = make_bin_table(uni.distance)
        model = StockGraphormer(num_class=P + 1, num_atoms=P + 1, bias_dtype=torch.bfloat16 if bf16 else torch.float32,
                                act_dtype=torch.bfloat16 if (bf16 and gemm_dtype == "bf16") else torch.float32,
                                fused_layers=fused, **args).to(device)
        coll = DeviceCollator(device, bin_table=table, multi_hop_max_dist=20, rel_pos_max=1024)
        return uni, model, coll
    kw = dict(bias_dtype=torch.bfloat16 if bf16 else torch.float32, gcn_dtype=torch.bfloat16 if bf16 else torch.float32,
              act_dtype=torch.bfloat16 if (bf16 and gemm_dtype == "bf16") else torch.float32, fused_layers=fused)
    if name == "big":
        uni = synth.make_sparse_universe(P=P, n_cat=w["n_cat"], n_user=w["n_user"], seed=seed)
        model = Graphormer(universe=uni, num_bins=uni.num_bins + 2, **kw, **args).to(device)
        coll = DeviceCollator(device, coords=uni.coords, bin_edges=uni.bin_edges, multi_hop_max_dist=20, rel_pos_max=1024)
    else:
        uni = synth.make_universe(P=P, n_cat=w["n_cat"], n_user=w["n_user"], seed=seed)
        num_bins, _, table = make_bin_table(uni.distance)
        model = Graphormer(universe=uni, num_bins=num_bins + 2, **kw, **args).to(device)
        coll = DeviceCollator(device, bin_table=table, multi_hop_max_dist=20, rel_pos_max=1024)
    return uni, model, coll


def describe(name, P=None, variant="fq"):
    w = WORKLOADS[name]
    m = w["model"]
    if variant == "stock":
        return ("%s trajectories on model.py's Graphormer (pre-LN EncoderLayer), P=%d, hidden %d (C=%d, d=%d), %d layers, %d heads, "
                "ffn %d, multi_hop_max_dist %d, dropout %.1f, (P+1)-way head, fwd+cross_entropy+bwd+allreduce+AdamW"
                % (w["label"], int(P or w["P"]), m["hidden_dim"], m["hidden_dim"], m["hidden_dim"] // m["num_heads"], m["n_layers"],
                   m["num_heads"], m["ffn_dim"], m["multi_hop_max_dist"], m["dropout_rate"]))
    C = m["hidden_dim"] + 64
    return ("%s: model_fqandtoyo Graphormer, %s, P=%d, hidden %d (C=%d, d=%d), %d layers, %d heads, ffn %d, "
            "multi_hop_max_dist %d, dropout %.1f, fwd+GradientTailLoss+bwd+allreduce+AdamW"
            % (w["label"], m["dataset_name"], int(P or w["P"]), m["hidden_dim"], C, C // m["num_heads"], m["n_layers"],
               m["num_heads"], m["ffn_dim"], m["multi_hop_max_dist"], m["dropout_rate"]))
