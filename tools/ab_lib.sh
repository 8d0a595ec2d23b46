#!/bin/bash
# A/B of two library builds on one box: build another revision's csrc/ to ag2video_amd/csrc/libag2v_prev.so first, e.g.
#   git archive <rev> ag2video_amd/csrc | tar -x -C /tmp/prev && make -C /tmp/prev/ag2video_amd/csrc && cp /tmp/prev/ag2video_amd/csrc/libag2v.so ag2video_amd/csrc/libag2v_prev.so
for i in 1 2 3; do
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['ms_per_step'])"
  AG2V_LIB_PATH=$PWD/ag2video_amd/csrc/libag2v_prev.so python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['ms_per_step'])"
done
