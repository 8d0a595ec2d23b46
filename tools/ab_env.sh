#!/bin/bash
# A/B of one environment switch on the c3 step: tools/ab_env.sh VAR [reps]
V=$1; R=${2:-2}
for i in $(seq 1 $R); do
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default      ', d['ms_per_step'], d['extra']['peak_mem_gb'])"
  env $V=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', d['ms_per_step'], d['extra']['peak_mem_gb'])"
done
