import sys, collections, torch
sys.path.insert(0, '.')
import bench
from ag2video_amd import ops, _lib
cnt = collections.Counter()
orig = _lib.call
def call(name, *a):
    if name == 'ag2v_gemm_f32':
        M, N, K, sam, sak, sbk, sbn = a[4], a[5], a[6], a[7], a[8], a[9], a[10]
        pa = a[0].value if hasattr(a[0], 'value') else int(a[0] or 0); pb = a[1].value if hasattr(a[1], 'value') else int(a[1] or 0)
        ok = (K % 4 == 0 and K <= 1152 and -(-N // 32) * -(-M // 32) <= 1024 and pa % 16 == 0 and pb % 16 == 0 and (sak != 1 or sam % 4 == 0) and (sbk != 1 or sbn % 4 == 0))
        cnt[(M, N, K, a[13], 'oneshot' if ok else 'loop pa%%16=%d pb%%16=%d sam=%d sbn=%d' % (pa % 16, pb % 16, sam, sbn))] += 1
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
