#!/bin/bash
# Clock, matrix-pipe and wait counters of the halo-tile convolution kernels against the kernels they replace, on one shape.
# usage (GPU box, repo root): bash tools/pmc_halo.sh [N,C,H,W,K]     -> gpurun_out/pmc_halo_<shape>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-31,512,256,256,128}
rm -rf /tmp/ph
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/ph -- python3 $R/tools/halo_bench.py 3 $S > /tmp/ph.log 2>&1
python3 - "$S" <<'PY' | tee $R/gpurun_out/pmc_halo_${S//,/_}.txt
import csv, glob, sys, collections
shape = [int(v) for v in sys.argv[1].split(',')]
N, C, H, W, K = shape
mfma_cycles = N * K * H * W * C * 9 * 3 / 16384.0 * 32          # matrix-pipe cycles the launch needs (32 per v_mfma_f32_32x32x16_bf16)
fc = glob.glob('/tmp/ph/*/*counter_collection.csv'); ft = glob.glob('/tmp/ph/*/*kernel_trace.csv')
dur = collections.defaultdict(list)
for r in csv.DictReader(open(ft[0])):
    n = r['Kernel_Name']
    if 'conv_halo' in n or 'conv_igemm_bf16x3_ps' in n:
        dur[n.split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(fc[0])):
    n = r['Kernel_Name']
    if 'conv_halo' in n or 'conv_igemm_bf16x3_ps' in n:
        k = n.split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k] += r['Counter_Name'] == 'GRBM_GUI_ACTIVE'
print('shape', shape, ' matrix-pipe cycles needed per SIMD: %.3e' % (mfma_cycles / 1024))
for k in agg:
    a, n = agg[k], cnt[k]
    d = sorted(dur[k])[len(dur[k]) // 2]
    cyc = a['GRBM_GUI_ACTIVE'] / 8 / n                      # per-XCD active cycles of one launch
    print('%s\n   launches %d  median %.3f ms under the profiler  clock %.2f GHz  matrix pipe busy (needed / elapsed cycles) %.1f %%  counter MFMA_BUSY / (4 SIMD x 256 CU x cycles) %.1f %%'
          % (k[:90], n, d * 1e3, cyc / d / 1e9, 100 * mfma_cycles / 1024 / cyc, 100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / n / (1024 * cyc)))
    wc = a['SQ_WAVE_CYCLES']
    print('   of wave cycles: waiting (s_waitcnt / barrier) %.1f %%, issue stalls %.1f %%, issuing %.1f %%;  LDS active %.1f %% of CU cycles, bank conflicts %.1f %% of that'
          % (100 * a['SQ_WAIT_ANY'] / wc, 100 * a['SQ_WAIT_INST_ANY'] / wc, 100 * a['SQ_ACTIVE_INST_ANY'] / wc,
             100 * a['SQ_LDS_IDX_ACTIVE'] / n / (cyc * 256), 100 * a['SQ_LDS_BANK_CONFLICT'] / max(1.0, a['SQ_LDS_IDX_ACTIVE'])))
PY
