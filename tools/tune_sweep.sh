#!/bin/bash
# Developer tool: bench.py under several TunableOp measurement settings (two runs each).
for e in "X=1" "PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE=0" "PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE=0 PYTORCH_TUNABLEOP_ICACHE_FLUSH_ENABLED=0" "PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE=0 PYTORCH_TUNABLEOP_ICACHE_FLUSH_ENABLED=0 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=100" "PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=100"; do
  echo "$e"
  for r in 1 2; do
    t0=$(date +%s)
    env $e python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['final_loss'])"
    echo "  wall $(( $(date +%s) - t0 )) s"
  done
done
