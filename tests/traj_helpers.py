"""Shared by the G10 tests (CPU oracle and GPU trainer): the sample of a parameter the fixture keeps
(tests/golden/make_golden_traj.py:param_sample)."""
import numpy as np


def param_sample(p):
    f = np.asarray(p).reshape(-1)
    return f if f.size <= 512 else f[np.linspace(0, f.size - 1, 512).astype(np.int64)]
