#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the bf16x3 conv kernels on tools/pmc_traffic.py's shapes (two separate PMC passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-pmc_quick}
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_fetch -- python3 $R/tools/pmc_traffic.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_write -- python3 $R/tools/pmc_traffic.py > /dev/null 2>&1
python3 - $R/gpurun_out/${T} <<'PY'
import csv, glob, collections, sys
def agg(path, counter):
    disp = collections.defaultdict(float); name = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            disp[r['Dispatch_Id']] += float(r['Counter_Value']); name[r['Dispatch_Id']] = r['Kernel_Name']
    per = collections.defaultdict(list)
    for d, v in disp.items(): per[name[d]].append(v)
    return per
f = agg(glob.glob(sys.argv[1] + '_fetch/*/*counter_collection.csv')[0], 'FETCH_SIZE')
w = agg(glob.glob(sys.argv[1] + '_write/*/*counter_collection.csv')[0], 'WRITE_SIZE')
for k in f:
    if 'bf16x3' in k:
        print(k[:72], '| FETCH MB (x2 corrected)', [round(x * 2 * 1.024e-3) for x in f[k]], '| WRITE MB', [round(x * 1.024e-3) for x in w.get(k, [])])
PY
