import sys, collections, torch
sys.path.insert(0, '.')
import bench
from ag2video_amd import ops, _lib
cnt = collections.Counter()
orig = _lib.call
def call(name, *a):
    if name == 'ag2v_gemm_f32':
        cnt[(a[4], a[5], a[6], a[13])] += 1     # M, N, K, accumulate
    return orig(name, *a)
_lib.call = call
ops._lib.call = call
import argparse
sys.argv = ['bench.py', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-fp32-leg', '--no-kernel-timing']
try:
    bench.main()
except SystemExit:
    pass
tot = sum(cnt.values())
print('gemm calls', tot)
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:30]:
    print(k, v)
