#!/bin/bash
# Issue / wait / LDS / VALU counters of the layout-convolution kernels on the c3 shape (tools/lc_bench.py, tools/lc_bwd_bench.py).
# usage (GPU box, repo root): [AG2V_NO_LC_FWD3=1] bash tools/pmc_lc.sh [tag]     -> gpurun_out/pmc_layout_conv[_tag].txt
# (AG2V_NO_LC_FWD3=1: the second form, layout_conv_fwd_kernel; default: layout_conv_fwd3_kernel.)  The four dispatches printed per pass are
# lc_bench's last ones: dense with / without image channels, sparse (6 of 22 slots: c3's occupancy) with / without.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pl1 /tmp/pl2
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/pl1 -- python3 $R/tools/lc_bench.py > /tmp/pl1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR \
  --kernel-trace --output-format csv -d /tmp/pl2 -- python3 $R/tools/lc_bench.py > /tmp/pl2.log 2>&1
python3 - <<'PY' | tee $R/gpurun_out/pmc_layout_conv${1:+_$1}.txt
import csv, glob, collections
for d in ('/tmp/pl1', '/tmp/pl2'):
    fc = glob.glob(d + '/*/*counter_collection.csv'); ft = glob.glob(d + '/*/*kernel_trace.csv')
    if not fc:
        print(d, 'no counters'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fc[0])):
        n = r['Kernel_Name']
        if 'layout_conv' in n:
            agg[n.split('(')[0] + ' #' + r['Dispatch_Id']][r['Counter_Name']].append(float(r['Counter_Value']))
    durs = {}
    for r in csv.DictReader(open(ft[0])):
        if 'layout_conv' in r['Kernel_Name']:
            durs[r['Kernel_Name'].split('(')[0] + ' #' + r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
    for k in list(agg)[-4:]:
        a = {c: sum(v) for c, v in agg[k].items()}
        print(k, '%.2f ms' % durs.get(k, -1), {c: '%.3e' % v for c, v in a.items()})
PY
