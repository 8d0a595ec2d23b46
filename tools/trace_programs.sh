# usage (GPU box, repo root): bash tools/trace_programs.sh
# Kernel trace of 3 c3 steps: when do the graph-net programs of a step run, on which queue, and what overlaps them?
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -- python3 $R/bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fp32-leg > /tmp/tp.log 2>&1
cd $R
python3 - <<'PY' | tee gpurun_out/trace_programs.txt
import csv, glob
f = glob.glob('/tmp/tp/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
progs = [r for r in rows if r[3].startswith('graph_program_kernel')]
t0 = rows[0][0]
print('%d dispatches, %d graph programs' % (len(rows), len(progs)))
for s, e, q, n in progs[-12:]:
    # busy time of OTHER kernels inside this program's window
    other = 0
    names = {}
    for s2, e2, q2, n2 in rows:
        if e2 <= s or s2 >= e or (s2 == s and e2 == e and q2 == q):
            continue
        o = min(e, e2) - max(s, s2)
        other += o
        k = n2.split('(')[0][:48]
        names[k] = names.get(k, 0) + o
    top = sorted(names.items(), key=lambda kv: -kv[1])[:3]
    print('program  start %9.3f ms  dur %6.3f ms  queue %s  other kernels inside its window: %6.3f ms  %s'
          % ((s - t0) / 1e6, (e - s) / 1e6, q, other / 1e6, [(k, round(v / 1e6, 2)) for k, v in top]))
PY
