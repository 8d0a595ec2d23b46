#!/bin/bash
# LDS bank-conflict counters + time of igemm tile variants.  Build them first (no-stamps builds of igemm_stamps.hip):
#   for v in "40 true" "48 true" "40 false" "48 false"; do set -- $v; hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNO_STAMPS \
#       -DAG2V_LDK=$1 -DAG2V_TALL_MF16=$2 -o tools/ubench/igemm_ns_$1_$2 tools/ubench/igemm_stamps.hip; done
# (AG2V_TALL_MF16 selects the MFMA shape of the table-path 256x128 tile; run the binaries without TAPK=1 to see it)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 40_true 48_true 40_false 48_false; do
  for shape in "512 128" "128 128"; do
    rm -rf /tmp/pv; export TAPK=1
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pv -- $R/tools/ubench/igemm_ns_$v $shape > /tmp/pv.log 2>&1
    T=$(grep "tile:" /tmp/pv.log)
    python3 - "$v" "$shape" "$T" <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/pv/*/*counter_collection.csv')
agg = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(f[0])):
    if 'igemm' in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value'])
        n += r['Counter_Name'] == 'GRBM_GUI_ACTIVE'
print(sys.argv[1], sys.argv[2], '| launches', n, '| LDS active %.1f%% of CU cycles, conflict %.1f%% of active |' % (
    100 * agg['SQ_LDS_IDX_ACTIVE'] / (agg['GRBM_GUI_ACTIVE'] / 8 * 256), 100 * agg['SQ_LDS_BANK_CONFLICT'] / max(1, agg['SQ_LDS_IDX_ACTIVE'])), sys.argv[3][:60])
PY
  done
done
