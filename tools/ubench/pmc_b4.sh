cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in 0 1; do
  rm -rf /tmp/pv; export TAPK=1
  if [ $e = 1 ]; then export AG2V_NO_B4=1; else unset AG2V_NO_B4; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pv -- $R/tools/ubench/ns_b4 128 512 > /tmp/pv.log 2>&1
  python3 - $e <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/pv/*/*counter_collection.csv')
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f[0])):
    if 'igemm' in r['Kernel_Name']: agg[r['Counter_Name']] += float(r['Counter_Value'])
print('NO_B4=%s' % sys.argv[1], 'LDS active %.1f%% of CU cycles, conflict %.1f%% of active' % (100 * agg['SQ_LDS_IDX_ACTIVE'] / (agg['GRBM_GUI_ACTIVE'] / 8 * 256), 100 * agg['SQ_LDS_BANK_CONFLICT'] / max(1, agg['SQ_LDS_IDX_ACTIVE'])))
PY
done
