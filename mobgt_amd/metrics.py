"""Evaluation metrics of `graphormer/model_fqandtoyo.py`: `get_acc` (:48-90) and
`MRR_metric` (:122-131), evaluated on the device with one `topk` / one rank computation for the whole
batch instead of per-row Python loops.  Quirks kept: row order of the result (`[top10, top5, top1, top20]`),
targets are class ids already shifted by the caller (`y - 1`, :1487), and `get_acc` stops at the FIRST row
whose target is 0 (:88-89) -- rows after it are ignored, exactly as in the reference.
"""
import numpy as np
import torch


def get_acc(target, scores):
    """-> (acc [4,1], ndcg [4,1]) numpy arrays: rows = top-10, top-5, top-1, top-20 hit counts / DCG sums."""
    target = torch.as_tensor(target).reshape(-1).to(scores.device)
    n = target.numel()
    zero = (target == 0).nonzero()
    stop = int(zero[0]) if zero.numel() else n                       # `else: break` at the first target == 0
    acc, ndcg = np.zeros((4, 1)), np.zeros((4, 1))
    if stop == 0:
        return acc, ndcg
    if scores.is_cuda:                                                 # one rank kernel instead of a top-k + search
        from . import ops
        rank = ops.target_rank(scores[:stop], target[:stop])[:, 0].long()
        found = (rank >= 0) & (rank < 20)
    else:
        _, idx = scores[:stop].topk(20, dim=1)
        hit = idx == target[:stop].unsqueeze(1)                        # at most one True per row
        rank = hit.float().argmax(dim=1)                               # position of the hit (0 when no hit)
        found = hit.any(dim=1)
    gain = 1.0 / torch.log2(rank.clamp(min=0).double() + 2.0)
    for row, k in ((3, 20), (0, 10), (1, 5), (2, 1)):
        m = found & (rank < k)
        acc[row] = float(m.sum())
        ndcg[row] = float(gain[m].sum())
    return acc, ndcg


def MRR_metric(target, scores):
    """Sum over rows of 1 / rank of the target under a descending sort (ties: numpy argsort order of the
    reference is approximated by counting strictly greater scores plus earlier-index... see note)."""
    target = torch.as_tensor(target).reshape(-1).to(scores.device)
    if scores.is_cuda:
        from . import ops
        r_idx = ops.target_rank(scores, target)[:, 1]
        return float((1.0 / (r_idx.double() + 1.0)).sum())
    s = scores.double()
    t = s.gather(1, target.long().unsqueeze(1))
    # reference: rec_list = argsort(row)[::-1]; r_idx = position of the target.  For distinct scores this is the
    # number of strictly larger scores; with ties, reversed ascending argsort puts LATER indices first.
    greater = (s > t).sum(dim=1)
    cols = torch.arange(s.shape[1], device=s.device).unsqueeze(0)
    ties_before = ((s == t) & (cols > target.long().unsqueeze(1))).sum(dim=1)
    r_idx = greater + ties_before
    return float((1.0 / (r_idx.double() + 1.0)).sum())


def evaluate_outputs(outputs):
    """`test_epoch_end` bookkeeping (model_fqandtoyo.py:1546-1597) over a list of {"y_pred": [poi, cat], "y_true"}:
    returns dict(acc@1/5/10/20, ndcg@1/5/10/20, mrr), each averaged over the number of test samples."""
    tot = np.zeros(8)
    mrr, n = 0.0, 0
    for o in outputs:
        y_pred, y_true = o["y_pred"][0], o["y_true"]
        a, d = get_acc(y_true, y_pred)
        tot += np.array([a[2, 0], a[1, 0], a[0, 0], d[2, 0], d[1, 0], d[0, 0], a[3, 0], d[3, 0]])
        mrr += MRR_metric(y_true, y_pred)
        n += len(y_true)
    tot /= max(n, 1)
    return {"acc@1": tot[0], "acc@5": tot[1], "acc@10": tot[2], "ndcg@1": tot[3], "ndcg@5": tot[4], "ndcg@10": tot[5],
            "acc@20": tot[6], "ndcg@20": tot[7], "mrr": mrr / max(n, 1)}
