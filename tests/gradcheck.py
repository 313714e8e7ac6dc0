"""Shared gradient comparison of the parity tests: ELEMENTWISE, tolerance scaled by the RMS of the reference's non-zero
entries (not by its maximum, and not a norm -- a transposed, permuted or mis-scaled gradient must fail)."""
import numpy as np


def grad_sample(g):
    """The part of a parameter gradient the golden fixtures keep (tests/golden/make_golden_model.py:grad_sample)."""
    g = np.asarray(g)
    return g if g.size <= 4096 or g.ndim < 2 else g[::7]


def grad_errors(got, ref):
    """(rms of the non-zero reference entries, relative L2 error, 99.9 % quantile and maximum of
    |err| / (0.15 rms + 0.05 |ref|) over those entries, largest |got| where the reference is exactly 0)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    nz = ref != 0
    rms = float(np.sqrt((ref[nz] ** 2).mean())) if nz.any() else 0.0
    rel_l2 = float(np.sqrt(((got - ref) ** 2).sum()) / max(np.sqrt((ref ** 2).sum()), 1e-30))
    ratio = np.abs(got - ref)[nz] / (0.15 * rms + 0.05 * np.abs(ref[nz])) if nz.any() else np.zeros(1)
    # (fewer than 2000 entries: the 99.9 % bound does not apply, only the 4x bound on the maximum)
    q999 = float(np.quantile(ratio, 0.999)) if ratio.size >= 2000 else min(float(ratio.max()), 1.0)
    stray = float(np.abs(got[~nz]).max()) if (~nz).any() else 0.0
    return rms, rel_l2, q999, float(ratio.max()), stray


def assert_grad_close(got, ref, name, max_rel_l2=4e-2, kink=4.0, scale=1.0):
    """`scale` tightens (< 1) or loosens the elementwise bound; `kink`: allowance for the last 0.1 % of the entries
    (derivative kinks of LeakyReLU / ELU, see tests/test_gpu_bench_parity.py)."""
    rms, rel_l2, q999, mx, stray = grad_errors(got, ref)
    assert rel_l2 <= max_rel_l2, (name, "relative L2", rel_l2)
    assert q999 <= scale and mx <= kink * scale, (name, "elementwise", q999, mx, rms)
    assert stray <= 1e-3 * rms + 1e-12, (name, "non-zero where the reference is exactly zero", stray)
    return rel_l2


# ---------------------------------------------------------------------------------------------- LeakyReLU branch replay at the head
# The classifier head (model_fqandtoyo.py:1353-1364) runs FuseEmbeddings -- Linear + LeakyReLU(0.2) -- on the G graph-token rows
# only (16 x 384 units at bench sizes).  A pre-activation within the forward pass's round-off (1e-3 with bf16 operands) of zero
# takes the other branch on one side, and ONE such unit moves the whole backward pass by ~1 % relative L2, two or three by 2-4 %
# (measured on the CPU by emulating the kernels' rounding points inside the oracle: with the branch pattern held fixed the same
# arithmetic is 0.2-0.6 % from fp32 on every parameter).  Like the dropout masks, the pattern is therefore
# REPLAYED: taken from the device run, imposed on the oracle -- the comparison is then between two evaluations of the same
# piecewise-linear branch of the network, and the gates can be as tight as the arithmetic is.
def device_head_pattern(model, batch, enc_out=None, state=None):
    """[G, W] bool (cpu): which units of embed_fuse_model3 are on the positive side in the device's forward of `batch` (taken from
    the encoder output the device produced, in fp32: model._enc_out after a forward here, or `enc_out` = the graph-token rows a
    replayed step graph copied aside, train.TrainStep(keep_head_rows=True).enc_outs[i]), and the pre-activations themselves.
    `state`: the state dict the step STARTED from (a replayed train step has already moved the model's parameters)."""
    import torch
    with torch.no_grad():
        if enc_out is None:
            model(batch)
            enc_out = model._enc_out[:, 0, :]
        enc = enc_out.float()                  # [G, C]: the graph-token rows
        sd = state if state is not None else model.state_dict()
        w = lambda k: sd[k].to(enc.device).float()
        user = w("user_embed_model.user_embedding.weight")[batch.user.long().view(-1) - 1]
        pre = torch.cat([enc, user], 1) @ w("embed_fuse_model3.fuse_embed.weight").t() + w("embed_fuse_model3.fuse_embed.bias")
    return (pre > 0).cpu(), pre.cpu()


def replay_head(pattern, seen=None, pre_dev=None):
    """`act` hook for oracle.model_oracle.graphormer_fq_forward: LeakyReLU(0.2) of the head's graph-token rows with the branch of
    every unit taken from `pattern`; `seen` (dict) collects, per graph, how many units the oracle alone would have put on the other
    side and -- key ("pre", g) -- the largest |pre-activation| among them on the oracle's side and (with `pre_dev`, the device's
    pre-activations from device_head_pattern) on the device's: a replayed unit must sit within the forward's round-off of the
    kink on BOTH sides, or the replay would follow a wrong device value instead of catching it (assert_replay_bounded)."""
    import torch

    def act(site, pre):
        if not (isinstance(site, tuple) and site[0] == "embed_fuse_model3" and site[2] == 0):
            return None
        m = pattern[site[1]]
        if seen is not None:
            flipped = (pre.detach() > 0) != m
            seen[site[1]] = int(flipped.sum())
            big = float(pre.detach().abs()[flipped].max()) if bool(flipped.any()) else 0.0
            if pre_dev is not None and bool(flipped.any()):
                big = max(big, float(pre_dev[site[1]].abs()[flipped].max()))
            seen[("pre", site[1])] = big
        return torch.where(m, pre, 0.2 * pre)
    return act


MAX_FLIPPED, MAX_FLIPPED_PRE = 16, 5e-3


def n_flipped(seen):
    return sum(v for k, v in seen.items() if not isinstance(k, tuple))


def assert_replay_bounded(seen, max_units=MAX_FLIPPED, max_pre=MAX_FLIPPED_PRE, pre_dev=None):
    """The replay may only decide units that are UNDECIDED within the forward's round-off: at most `max_units` of the G x W units of
    a batch differ between the oracle's own forward and the device's, and each of them has |pre-activation| <= `max_pre` on both
    sides (bf16 operands through six layers: the forward's relative error is ~1 % of the values' scale -- 5e-3 absolute on the
    default-initialised models, whose pre-activations have rms ~0.3; with `pre_dev` given the bound is max(max_pre, 2 % of the
    pre-activations' rms), for models with other weight scales: golden G8's seeded weights).  A device pre-activation that is
    wrong by more than that fails here instead of being followed."""
    n = n_flipped(seen)
    worst = max([v for k, v in seen.items() if isinstance(k, tuple)] or [0.0])
    if pre_dev is not None:
        max_pre = max(max_pre, 2e-2 * float(pre_dev.double().pow(2).mean().sqrt()))
    assert n <= max_units, ("LeakyReLU replay: too many head units differ between device and oracle", n)
    assert worst <= max_pre, ("LeakyReLU replay: a replayed unit is not within round-off of the kink", worst, max_pre)
    return n, worst
