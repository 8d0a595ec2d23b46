#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the dominant kernels on tools/pmc_traffic.py's shapes: two separate PMC passes per case
# (counters are collected in their own runs, with --kernel-trace only), summarised into gpurun_out/<tag>/traffic.json by
# tools/pmc_summary.py.   usage (GPU box, repo root): bash tools/pmc_quick.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-pmc}
mkdir -p $R/gpurun_out/$T
for c in ${PMC_CASES:-mlp_shared_fwd mlp_shared_dgrad_merged mlp_shared_wgrad_merged conv_256_wgrad_128px mlp_shared_fwd_ps mlp_shared_dgrad_merged_ps mlp_shared_wgrad_merged_ps conv_256_wgrad_128px_ps layout_conv flow_res_fp32 norm spade_pair bn_pyramid layout_conv_bwd flow_res_h16 mlp_shared_wgrad_merged_fp32 gamma_beta_wgrad_ps gamma_beta_dgrad_ps gamma_beta_fwd_ps}; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/$T/tmp -- python3 $R/tools/pmc_traffic.py $c > $R/gpurun_out/$T/$c.$ctr.log 2>&1
    # keep only the rows of this repo's kernels (the torch RNG kernels' names are kilobytes long)
    f=$(ls $R/gpurun_out/$T/tmp/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] || { echo "no counter file for $c $ctr"; rm -rf $R/gpurun_out/$T/tmp; continue; }
    (head -1 $f; grep -E '"(void )?(conv_|norm_|bn_stats|spade_pair|layout_conv)' $f) > $R/gpurun_out/$T/${c}_${ctr}_counter_collection.csv
    rm -rf $R/gpurun_out/$T/tmp
  done
done
cd $R
python3 tools/pmc_summary.py gpurun_out/$T
