#!/bin/bash
# Developer tool (GPU box): S-BIG step with the current library vs mobgt_amd/libmobgt_hip_base.so, interleaved.
cd "$GRAFT_REPO_ROOT"
for which in base new base new; do
  if [ $which = base ]; then export MOBGT_HIP_LIB=$GRAFT_REPO_ROOT/mobgt_amd/libmobgt_hip_base.so; else unset MOBGT_HIP_LIB; fi
  python bench.py --workload big --no-cpu-baseline --no-stress --no-parity --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which', round(d['value'],1), round(d['ms_per_step'],3), d['final_loss'])"
done
