"""ORACLE (test infrastructure): op-for-op CPU restatement, in plain torch fp32, of the model side of
the hot path -- `graphormer/model.py` (stock Graphormer) and `graphormer/model_fqandtoyo.py` (the
"fq" variant `entry.py` runs).  Functional style: every function takes a state dict `sd`
(reference parameter names -> tensors) so that reference checkpoints / the seeded fixtures plug
straight in, and autograd gives the backward.

Pinned against tests/golden/g4_encoder.npz, g5_bias.npz, g6_e2e.npz, g7_lr_loss.npz -- outputs of
the reference itself (tests/golden/make_golden*.py).  Also timed as bench.py's `cpu_baseline`
(kind "port"): the per-sample Python loops of the fq forward are kept as the reference has them.

The product (mobgt_amd/) never imports this module.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------ encoder layer
def linear(sd, prefix, x):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def layer_norm(sd, prefix, x):
    w = sd[prefix + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[prefix + ".bias"], 1e-5)


def _dropout(x, p, training, drop=None, site=None):
    """nn.Dropout / F.dropout.  `drop`: optional callable (x, p, site) -> tensor standing in for torch's generator in
    training mode -- the parity tests pass the keep masks the device drew (mobgt_*_mask_host), so that a training-mode step
    of the HIP path can be compared element by element; `site` names the call (see the callers)."""
    if drop is not None and training and p > 0:
        return drop(x, p, site)
    return F.dropout(x, p, training)


def multi_head_attention(sd, prefix, q, k, v, attn_bias, num_heads, p_drop=0.0, training=False, drop=None, mask=None):
    """model.py:424-460 / model_fqandtoyo.py:1675-1711.  `mask` [B, q_len, k_len] bool (:446-448; every caller of the
    reference passes None): the score of a masked pair -- bias included -- becomes 0, not -inf."""
    orig = q.size()
    B = q.size(0)
    d = sd[prefix + ".linear_q.weight"].shape[0] // num_heads
    scale = d ** -0.5
    q = linear(sd, prefix + ".linear_q", q).view(B, -1, num_heads, d).transpose(1, 2)
    k = linear(sd, prefix + ".linear_k", k).view(B, -1, num_heads, d).transpose(1, 2).transpose(2, 3)
    v = linear(sd, prefix + ".linear_v", v).view(B, -1, num_heads, d).transpose(1, 2)
    q = q * scale                                   # scaled BEFORE the matmul (:442)
    x = torch.matmul(q, k)
    if attn_bias is not None:
        x = x + attn_bias                           # unscaled bias (:445)
    if mask is not None:
        x = x.masked_fill(mask.unsqueeze(1), 0)     # (:446-448)
    x = torch.softmax(x, dim=3)
    x = _dropout(x, p_drop, training, drop, (prefix, "att"))
    x = x.matmul(v)
    x = x.transpose(1, 2).contiguous().view(B, -1, num_heads * d)
    x = linear(sd, prefix + ".output_layer", x)
    assert x.size() == orig
    return x


def feed_forward(sd, prefix, x):
    """model.py:393-405: Linear -> exact-erf GELU -> Linear (dropout_rate unused)."""
    return linear(sd, prefix + ".layer2", F.gelu(linear(sd, prefix + ".layer1", x)))


def encoder_layer_stock(sd, prefix, x, attn_bias, num_heads, p=0.0, p_att=0.0, training=False, drop=None, mask=None):
    """model.py:479-489 (pre-LN)."""
    y = layer_norm(sd, prefix + ".self_attention_norm", x)
    y = multi_head_attention(sd, prefix + ".self_attention", y, y, y, attn_bias, num_heads, p_att, training, drop, mask)
    x = x + _dropout(y, p, training, drop, (prefix, "res1"))
    y = feed_forward(sd, prefix + ".ffn", layer_norm(sd, prefix + ".ffn_norm", x))
    return x + _dropout(y, p, training, drop, (prefix, "res2"))


def encoder_layer_fq(sd, prefix, x, attn_bias, num_heads, p=0.0, p_att=0.0, training=False, drop=None, mask=None):
    """model_fqandtoyo.py:1731-1743: no pre-norm on attention; LN1 before FFN; LN2 on the output."""
    y = multi_head_attention(sd, prefix + ".self_attention", x, x, x, attn_bias, num_heads, p_att, training, drop, mask)
    x = x + _dropout(y, p, training, drop, (prefix, "res1"))
    y = feed_forward(sd, prefix + ".ffn", layer_norm(sd, prefix + ".ffn_norm1", x))
    x = x + _dropout(y, p, training, drop, (prefix, "res2"))
    return layer_norm(sd, prefix + ".ffn_norm2", x)


# -------------------------------------------------------------------------------------- attn bias
def _edge_term(sd, rel_pos, edge_input, H, D, fp16_roundtrip):
    """model.py:157-182 / model_fqandtoyo.py:1168-1208 -> [G,H,N,N]."""
    n_graph, n_node = rel_pos.shape[:2]
    rel_pos_ = rel_pos.clone()
    rel_pos_[rel_pos_ == 0] = 1
    rel_pos_ = torch.where(rel_pos_ > 1, rel_pos_ - 1, rel_pos_)
    if D > 0:
        rel_pos_ = rel_pos_.clamp(0, D)
        edge_input = edge_input[:, :, :, :D, :]
    e = F.embedding(edge_input, sd["edge_encoder.weight"], padding_idx=0)     # [G,N,N,D,F,H]; nn.Embedding(padding_idx=0)
    if fp16_roundtrip:
        e = torch.cat([e[j].mean(-2).unsqueeze(0).half() for j in range(len(edge_input))], dim=0).float()
    else:
        e = e.mean(-2)
    max_dist = e.size(-2)
    flat = e.permute(3, 0, 1, 2, 4).reshape(max_dist, -1, H)
    W = sd["edge_dis_encoder.weight"]
    if fp16_roundtrip:
        flat = torch.bmm(flat.half(), W.half().reshape(-1, H, H)[:max_dist, :, :]).float()
    else:
        flat = torch.bmm(flat, W.reshape(-1, H, H)[:max_dist, :, :])
    e = flat.reshape(max_dist, n_graph, n_node, n_node, H).permute(1, 2, 3, 0, 4)
    return (e.sum(-2) / (rel_pos_.float().unsqueeze(-1))).permute(0, 3, 1, 2)


def assemble_bias(sd, batch, H, D, variant):
    """model.py:126-190 (variant 'stock') / model_fqandtoyo.py:1143-1216 (variant 'fq')."""
    attn_bias, rel_pos = batch.attn_bias, batch.rel_pos
    g = attn_bias.clone().unsqueeze(1).repeat(1, H, 1, 1)
    # (every table below is an nn.Embedding(..., padding_idx=0) in the reference, model.py:47-60 / model_fqandtoyo.py:
    # 638-642, 756-788: row 0 is read like any other row but never receives a gradient)
    rel = F.embedding(rel_pos, sd["rel_pos_encoder.weight"], padding_idx=0).permute(0, 3, 1, 2)
    if variant == "fq":
        rel = rel + F.embedding(batch.poi_pos, sd["poi_pos_encoder.weight"], padding_idx=0).permute(0, 3, 1, 2)
    g[:, :, 1:, 1:] = g[:, :, 1:, 1:] + rel
    t = sd["graph_token_virtual_distance.weight"].view(1, H, 1).unsqueeze(-2)
    g[:, :, 1:, :1] = g[:, :, 1:, :1] + t            # column 0 only; the row-0 add is commented out upstream
    g[:, :, 1:, 1:] = g[:, :, 1:, 1:] + _edge_term(sd, rel_pos, batch.edge_input, H, D, variant == "fq")
    return g + attn_bias.unsqueeze(1)                # attn_bias counted twice (:190)


# ------------------------------------------------------------------------------- stock Graphormer
def graphormer_stock_forward(sd, batch, n_layers, H, D, p=0.0, p_in=0.0, p_att=0.0, training=False, drop=None):
    """model.py:111-217 (non-PCQM branch: downstream_out_proj on the graph token)."""
    x = batch.x
    in_degree = out_degree = batch.in_degree         # model.py:118 aliases out_degree to in_degree
    n_graph = x.size(0)
    bias = assemble_bias(sd, batch, H, D, "stock")
    node = F.embedding(x, sd["atom_encoder.weight"], padding_idx=0).sum(dim=-2)
    node = node + F.embedding(in_degree, sd["in_degree_encoder.weight"], padding_idx=0) \
        + F.embedding(out_degree, sd["out_degree_encoder.weight"], padding_idx=0)
    tok = sd["graph_token.weight"].unsqueeze(0).repeat(n_graph, 1, 1)
    out = _dropout(torch.cat([tok, node], dim=1), p_in, training, drop, "input")
    for l in range(n_layers):
        out = encoder_layer_stock(sd, f"layers.{l}", out, bias, H, p, p_att, training, drop)
    out = layer_norm(sd, "final_ln", out)
    return linear(sd, "downstream_out_proj", out[:, 0, :])


# ------------------------------------------------------------------------------------ fq Graphormer
def calculate_laplacian_matrix(adj_mat, diag_inverse=False):
    """model_fqandtoyo.py:456-486, mat_type 'hat_rw_normd_lap_mat': (D+I)^-1 (A+I), row-sum degrees.
    `diag_inverse`: D+I is diagonal, so its inverse is the reciprocal of the diagonal -- the same matrix without the
    reference's O(P^3) `matrix_power(-1)` (a minute at P = 7856); tests/test_oracle_model.py pins the two against each
    other at P = 64."""
    adj = np.asarray(adj_mat, dtype=np.float64)
    n = adj.shape[0]
    if diag_inverse:
        return (adj + np.identity(n)) / (np.sum(adj, axis=1) + 1.0)[:, None]
    deg = np.diag(np.sum(adj, axis=1))
    return np.matmul(np.linalg.matrix_power(deg + np.identity(n), -1), adj + np.identity(n))


def freedman_diaconis_bins(x):
    iqr = np.subtract(*np.percentile(x, [75, 25]))
    binsize = 2 * iqr * np.power(len(x), -1 / 3)
    return int(np.ceil((np.max(x) - np.min(x)) / binsize))


def fq_constants(uni, dataset_name, diag_inverse=False, num_bins=None):
    """The non-trainable tensors `model_fqandtoyo.Graphormer.__init__` derives from Graph_*.csv and the
    distance pickle (:650-700 gowalla, :787-838 foursquaregraph).  `diag_inverse`: see calculate_laplacian_matrix;
    `num_bins`: skip the Freedman-Diaconis pass over the distance matrix (the caller knows the table size)."""
    raw_X = uni.poi_table
    cats = raw_X[:, 4]
    uniq = np.unique(cats)                               # OneHotEncoder category order = sorted unique
    num_cats = len(uniq)
    P = raw_X.shape[0]
    onehot = (cats[:, None] == uniq[None, :]).astype(np.float32)
    X = np.zeros((P, 3 + num_cats), dtype=np.float32)
    X[:, 0] = raw_X[:, 1]
    X[:, 1:num_cats + 1] = onehot
    X[:, num_cats + 1] = raw_X[:, 2]
    X[:, num_cats + 2] = raw_X[:, 3]
    C_X = (np.arange(1, num_cats + 1)[:, None] == uniq[None, :]).astype(np.float32)
    if num_bins is None:
        d = uni.distance
        if dataset_name == "foursquaregraph":
            dm = np.delete(d, 0, axis=0)                 # :893-894: only the row delete takes effect
        else:
            dm = np.delete(np.delete(d, 0, axis=0), 0, axis=1)
        num_bins = freedman_diaconis_bins(dm - dm.min())
    return SimpleNamespace(
        X=torch.from_numpy(X),
        D_A=torch.from_numpy(calculate_laplacian_matrix(uni.graph_dist, diag_inverse)).float(),
        C_X=torch.from_numpy(C_X),
        C_A=torch.from_numpy(calculate_laplacian_matrix(uni.graph_cat, diag_inverse)).float(),
        poi2cat={int(r[0]): int(r[4]) for r in raw_X},
        num_cats=num_cats, num_bins=num_bins, P=P,
    )


def gcn(sd, prefix, x, adj, p_drop, training, drop=None):
    """modelGNN.py:53-74 with GraphConvolution :21-50 (dense adj)."""
    n = len([k for k in sd if k.startswith(prefix + ".gcn.") and k.endswith(".weight")])
    for i in range(n - 1):
        x = F.leaky_relu(torch.mm(adj, torch.mm(x, sd[f"{prefix}.gcn.{i}.weight"])) + sd[f"{prefix}.gcn.{i}.bias"], 0.2)
    x = _dropout(x, p_drop, training, drop, (prefix, "gcn"))
    return torch.mm(adj, torch.mm(x, sd[f"{prefix}.gcn.{n - 1}.weight"])) + sd[f"{prefix}.gcn.{n - 1}.bias"]


def fuse(sd, prefix, a, b, act=None, site=None):
    """FuseEmbeddings, model_fqandtoyo.py:440-455: LeakyReLU_0.2(Linear(cat(a, b))).
    `act`: optional callable (site, pre-activation) -> activation, or None for "no opinion" -- the parity tests replay the
    device's LeakyReLU branch pattern at the classifier head (like `drop` replays its dropout masks): the head sees G rows
    only, so one pre-activation that a 1e-3 forward perturbation carries across zero moves the whole gradient by ~2 %
    (tests/test_gpu_bench_parity.py)."""
    pre = linear(sd, prefix + ".fuse_embed", torch.cat((a, b), dim=a.dim() - 1))
    if act is not None:
        y = act(site if site is not None else prefix, pre)
        if y is not None:
            return y
    return F.leaky_relu(pre, 0.2)


def graphormer_fq_forward(sd, batch, consts, n_layers, H, D, p=0.0, p_in=0.0, p_att=0.0, training=False,
                          hidden=128, time_dim=32, cat_dim=32, drop=None, act=None, dataset="foursquaregraph"):
    """model_fqandtoyo.py:1123-1432 for foursquaregraph / gowalla_*: returns (poi_logits, cat_logits).
    `dataset="toyotagraph"` (:902-1039, :1417-1428): the 48-slot time table is a plain nn.Embedding there (no padding_idx: row
    0 trains), and the POI head returns log-probabilities (F.log_softmax of the out_proj logits)."""
    x = batch.x
    G, N = x.size()[:2]
    C = hidden + time_dim + cat_dim
    bias = assemble_bias(sd, batch, H, D, "fq")
    indx = (x != 0).sum(dim=-2)                                                       # :1225-1228
    poidist = gcn(sd, "poi_distance_model", consts.X, consts.D_A, 0.3, training, drop)      # :1236
    catemb = gcn(sd, "poi_cat_model", consts.C_X, consts.C_A, 0.1, training, drop)          # :1237
    user_emb = torch.squeeze(F.embedding(batch.user - 1, sd["user_embed_model.user_embedding.weight"]))  # :1239-1240
    node_features = torch.zeros(G, N, C)
    for pi in range(G):                                                               # :1257-1269
        n = int(indx[pi][0])
        cat_e = catemb[torch.LongTensor([consts.poi2cat[int(x[pi][q])] - 1 for q in range(n)])]
        time_e = F.embedding((batch.time_normal[pi][:n] * 48).long(), sd["time_embed_model_48.weight"],
                             padding_idx=None if dataset == "toyotagraph" else 0).squeeze(1)
        poi_e = poidist[x[pi][:n] - 1].squeeze(1)
        f2 = fuse(sd, "embed_fuse_model2", poi_e, time_e)
        node_features[pi][:n] = fuse(sd, "embed_fuse_model4", f2, cat_e)
    # :1287-1298; poi_freq is all zero (:1243), i.e. the padding row of fre_embed_model: read, never trained
    node_features = node_features + F.embedding(torch.zeros(1, dtype=torch.long), sd["fre_embed_model.weight"], padding_idx=0)[0] \
        + F.embedding(batch.in_degree, sd["in_degree_encoder.weight"], padding_idx=0) \
        + F.embedding(batch.out_degree, sd["out_degree_encoder.weight"], padding_idx=0)
    pe = sd["pos_embed.pe"]
    nf = node_features.clone()
    for i in range(G):                                                                # :348-351 'node_reverse'
        n = int(indx[i][0])
        nf[i][:n] = nf[i][:n] + pe[1:n + 1]
    nf = _dropout(nf, 0.1, training, drop, "pos_nodes")                               # LearnablePositionalEncoding dropout :358
    tok = sd["graph_token.weight"].unsqueeze(0).repeat(G, 1, 1) + pe[0]               # :1338-1342 'pos0'
    tok = _dropout(tok, 0.1, training, drop, "pos_token")
    out = _dropout(torch.cat([tok, nf], dim=1), p_in, training, drop, "input")
    for l in range(n_layers):                                                         # :1347-1352
        out = encoder_layer_fq(sd, f"layers.{l}", out, bias, H, p, p_att, training, drop)
    rows = []
    for pi in range(G):                                                               # :1353-1358 (q over N, not N+1)
        rows.append(torch.stack([fuse(sd, "embed_fuse_model3", out[pi][q], user_emb[pi], act, ("embed_fuse_model3", pi, q)) for q in range(N)]))
    tmp = torch.stack(rows)
    o = _dropout(F.elu(layer_norm(sd, "final_ln", tmp)), p_in, training, drop, "output")   # :1360-1364
    poi = linear(sd, "out_proj", o[:, 0, :])
    if dataset == "toyotagraph":
        poi = F.log_softmax(poi, dim=1)                                               # :1417-1428
    return poi, linear(sd, "cat_decoder", o[:, 0, :])                                 # :1394-1396


def gradient_tail_loss(inputs, targets, alpha=0.25, beta=1, k=1):
    """model_fqandtoyo.py:545-550 with the hard-coded "cuda" replaced by the input's device."""
    one_hot = torch.zeros_like(inputs)
    one_hot.scatter_(1, targets[:len(inputs)].view(-1, 1), 1)
    prob = torch.sigmoid(inputs)
    loss = -alpha * (1 - prob) ** k * one_hot * torch.log(prob) - (1 - one_hot) * beta * prob ** k * torch.log(1 - prob)
    return loss.mean()


def fq_training_loss(sd, batch, consts, **kw):
    """model_fqandtoyo.py:1446-1460: y-1 targets, GradientTailLoss(alpha=0.2) on the POI head only."""
    logits, _ = graphormer_fq_forward(sd, batch, consts, **kw)
    return gradient_tail_loss(logits, batch.y - 1, 0.2)


def toyota_training_loss(sd, batch, consts, **kw):
    """model_fqandtoyo.py:1462-1471 (with :1262 for the category target): GradientTailLoss(category logits, category of the
    TARGET POI - 1, alpha = 0.1) + NLLLoss(ignore_index=0)(log-probabilities, y) -- y is NOT shifted here."""
    logp, cat_logits = graphormer_fq_forward(sd, batch, consts, dataset="toyotagraph", **kw)
    y = batch.y.view(-1)
    cat_target = torch.tensor([consts.poi2cat[int(v)] - 1 for v in y], dtype=torch.long)
    return gradient_tail_loss(cat_logits, cat_target, 0.1) + F.nll_loss(logp, y, ignore_index=0)


# ------------------------------------------------------------------------- schedule / metrics (§8f)
def polynomial_decay_lr(step_count, warmup_updates, tot_updates, lr, end_lr, power):
    """lr.py:17-31; `step_count` is the scheduler's _step_count (1 after construction)."""
    if step_count <= warmup_updates:
        return step_count / float(warmup_updates) * lr
    if step_count >= tot_updates:
        return end_lr
    pct = 1 - (step_count - warmup_updates) / (tot_updates - warmup_updates)
    return (lr - end_lr) * pct ** power + end_lr


def get_acc(target, scores):
    """model_fqandtoyo.py:48-90: rows [top10, top5, top1, top20]; stops at the first target == 0."""
    target = np.asarray(target)
    predx = torch.as_tensor(scores).topk(20, 1)[1].numpy()
    acc, ndcg = np.zeros((4, 1)), np.zeros((4, 1))
    for i, p in enumerate(predx):
        t = target[i]
        if t == 0:
            break
        for row, k in ((3, 20), (0, 10), (1, 5), (2, 1)):
            if t in p[:k]:
                acc[row] += 1
                ndcg[row] += 1.0 / np.log2(list(p[:k]).index(t) + 2)
    return acc, ndcg


def mrr_metric(target, scores):
    """model_fqandtoyo.py:122-131."""
    y_true, y_pred = np.asarray(target), np.asarray(scores)
    mrr = 0
    for j in range(len(y_pred)):
        rec = y_pred[j].argsort()[-len(y_pred[j]):][::-1]
        mrr += 1 / (np.where(rec == y_true[j])[0][0] + 1)
    return mrr
