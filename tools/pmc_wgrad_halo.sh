#!/bin/bash
# Wave-state / LDS / matrix-pipe counters of conv_wgrad_halo_bf16x3_kernel on the merged mlp_shared shape (separate --pmc passes).
# usage (GPU box, repo root): bash tools/pmc_wgrad_halo.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-pmc_wgrad_halo}
mkdir -p $R/gpurun_out/$T
export WGH_SHAPES=${WGH_SHAPES:-0} WGH_ONLY_NEW=1
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $ctrs | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/$T/tmp -- python3 $R/tools/wgrad_halo_bench.py 3 > $R/gpurun_out/$T/$tag.log 2>&1
  f=$(ls $R/gpurun_out/$T/tmp/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; grep -E 'conv_wgrad' $f) > $R/gpurun_out/$T/${tag}.csv
  k=$(ls $R/gpurun_out/$T/tmp/*/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$k" ] && (head -1 $k; grep -E 'conv_wgrad' $k) > $R/gpurun_out/$T/${tag}_trace.csv
  rm -rf $R/gpurun_out/$T/tmp
done
cd $R
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/$T/*.csv')):
    if f.endswith('_trace.csv'): continue
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        acc[r['Counter_Name']][0] += float(r['Counter_Value']); acc[r['Counter_Name']][1] += 1
    for k, (v, n) in acc.items():
        print('%-28s per launch %.4g (%d launches)' % (k, v / n, n))
PY
