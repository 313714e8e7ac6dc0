#!/bin/bash
# Developer tool (GPU box): S-BIG step, row-subset transposed spmm as gather vs atomic scatter (interleaved).
cd "$GRAFT_REPO_ROOT"
for cfg in 0 1 0 1; do
  MOBGT_SPMM_SCATTER=$cfg python bench.py --workload big --no-cpu-baseline --no-stress --no-parity --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('scatter=$cfg', round(d['value'],1), round(d['ms_per_step'],3), d['final_loss'])"
done
