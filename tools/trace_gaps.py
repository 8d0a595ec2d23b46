"""Idle time and cross-queue overlap in a rocprofv3 kernel trace.
usage: trace_gaps.py <kernel_trace.csv> [skip_fraction=0.3 | adam:<a>:<b>]
Looks at the part of the trace after `skip_fraction` of the dispatches (warm-up) -- or, with adam:a:b, at the window from the end
of the a-th adam_kernel dispatch to the end of the b-th (a gd_step launches two: generator, discriminator) -- and prints: wall, union-busy, idle, per-queue busy,
time with >= 2 queues busy, and the kernels that most often follow an idle gap (gap time attributed to the kernel that ends it)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
arg = sys.argv[2] if len(sys.argv) > 2 else '0.3'
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r['Kernel_Name']) for r in rows))
if arg.startswith('adam:'):
    a, b = [int(x) for x in arg.split(':')[1:]]
    ends = [e for s_, e, q, n in ev if n.startswith('adam_kernel')]
    lo, hi = ends[a - 1], ends[b - 1]
    ev = [x for x in ev if x[0] >= lo and x[1] <= hi]
else:
    ev = ev[int(len(ev) * float(arg)):]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = 0; cur_end = t0; gaps = collections.Counter(); gapn = collections.Counter(); idle = 0
for s, e, q, n in ev:
    if s > cur_end:
        g = s - cur_end
        idle += g; gaps[n[:70]] += g; gapn[n[:70]] += 1
    cur_end = max(cur_end, e)
wall = t1 - t0
perq = collections.Counter()
for s, e, q, n in ev: perq[q] += e - s
# overlap: sweep
pts = []
for s, e, q, n in ev: pts.append((s, 1)); pts.append((e, -1))
pts.sort(); depth = 0; last = pts[0][0]; ov = 0
for t, d in pts:
    if depth >= 2: ov += t - last
    depth += d; last = t
print('dispatches %d  wall %.1f ms  idle %.1f ms (%.1f %%)  >=2 kernels in flight %.1f ms' % (len(ev), wall / 1e6, idle / 1e6, 100.0 * idle / wall, ov / 1e6))
for q, b in perq.most_common(): print('  queue %s busy %.1f ms' % (q, b / 1e6))
print('idle time by the kernel that ends the gap:')
for n, g in gaps.most_common(25): print('  %8.2f ms %6d gaps  avg %6.1f us  %s' % (g / 1e6, gapn[n], g / gapn[n] / 1e3, n))
