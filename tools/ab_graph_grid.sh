#!/bin/bash
# c3 step time against the graph-net programs' grid (AG2V_GRAPH_GRID; default 128 workgroups) and with the side stream on
for i in 1 2; do
  for g in 128 96 64 48 32; do
    AG2V_GRAPH_GRID=$g python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid $g', d['ms_per_step'])"
  done
  AG2V_SIDE_STREAM=1 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side stream', d['ms_per_step'])"
done
