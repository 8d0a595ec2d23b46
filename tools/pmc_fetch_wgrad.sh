#!/bin/bash
# FETCH_SIZE per launch of the weight-gradient kernels (new and old) on WGH_CUSTOM shapes.  usage: bash tools/pmc_fetch_wgrad.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-pmc_fetch_wgrad}
mkdir -p $R/gpurun_out/$T
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$T/tmp -- python3 $R/tools/wgrad_halo_bench.py 2 > $R/gpurun_out/$T/fetch.log 2>&1
f=$(ls $R/gpurun_out/$T/tmp/*/*counter_collection.csv 2>/dev/null | head -1)
(head -1 $f; grep -E 'conv_wgrad' $f) > $R/gpurun_out/$T/fetch.csv
rm -rf $R/gpurun_out/$T/tmp
cd $R
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/$T/fetch.csv')))
acc = collections.OrderedDict()
for r in rows:
    k = (r['Kernel_Name'][:60], r['Grid_Size'])
    a = acc.setdefault(k, [0.0, 0]); a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in acc.items():
    print('%-62s grid %-8s FETCH %.3f GB per launch (%d launches)' % (k[0], k[1], 2 * 1024 * v / n / 1e9, n))
PY
