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
