#!/bin/bash
# Developer tool (GPU box): S-BIG step under different settings of the long-batch column sums.
cd "$GRAFT_REPO_ROOT"
for cfg in 64 128 256 512; do
  MOBGT_COLSUM_WIDE_WGS=$cfg python bench.py --workload big --no-cpu-baseline --no-stress --no-parity --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value'],1), round(d['ms_per_step'],3))"
done
