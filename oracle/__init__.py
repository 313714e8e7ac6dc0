"""ORACLE -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package; `mobgt_amd/` (the product) never does.  Each function cites the reference file:line it
follows.  Parity pinning: every function here is checked against the golden vectors in
`tests/golden/*.npz`, which were produced by running the reference itself in the build container
(`tests/golden/make_golden.py`); see `tests/test_oracle_*.py`.
"""
