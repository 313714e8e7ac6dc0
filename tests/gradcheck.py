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


def replay_head(pattern, seen=None):
    """`act` hook for oracle.model_oracle.graphormer_fq_forward: LeakyReLU(0.2) of the head's graph-token rows with the branch of
    every unit taken from `pattern`; `seen` (dict) collects how many units the oracle alone would have put on the other side."""
    import torch

    def act(site, pre):
        if not (isinstance(site, tuple) and site[0] == "embed_fuse_model3" and site[2] == 0):
            return None
        m = pattern[site[1]]
        if seen is not None:
            seen[site[1]] = int(((pre.detach() > 0) != m).sum())
        return torch.where(m, pre, 0.2 * pre)
    return act
