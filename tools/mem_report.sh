#!/bin/bash
# peak device memory and step time per configuration, default and lean tape -> gpurun_out/mem_report.txt
# usage (GPU box): bash tools/mem_report.sh
mkdir -p gpurun_out
out=gpurun_out/mem_report.txt
: > $out
for cfg in c1 c2 c3 c5; do
  for extra in "" "--lean-tape" ; do
    python bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-kernel-timing $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-4s %-12s %8.1f ms/step %7.1f frames/s  peak %6.1f GB' % ('$cfg', '$extra' or 'default', d['ms_per_step'], d['value'], d['extra']['peak_mem_gb']))" >> $out
  done
done
for extra in "" "--lean-tape" ; do
  python bench.py --config c5 --temporal-D --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-kernel-timing $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-4s %-12s %8.1f ms/step %7.1f frames/s  peak %6.1f GB' % ('c5+T', '$extra' or 'default', d['ms_per_step'], d['value'], d['extra']['peak_mem_gb']))" >> $out
done
cat $out
