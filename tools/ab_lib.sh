#!/bin/bash
# step time of two builds of the library, alternating in one lease: ab_lib.sh <other libag2v.so> [rounds] [config = c3]
# (AG2V_LIB_PATH selects the build, ag2video_amd/_lib.py)
other=$1; rounds=${2:-3}; cfg=${3:-c3}
line() { python bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for i in $(seq $rounds); do
  AG2V_LIB_PATH=$other line other
  line head
done
