# usage (GPU box, repo root): bash tools/evidence_r06_pmc.sh
# Round 6 counters: matrix-pipe / wait / LDS counters of the dominant forward kernel on the benched shapes (mlp_shared
# forward chunk, merged data gradient) and of conv_wgrad_halo<true, true>; FETCH / WRITE per launch of the split-word launches incl.
# the merged forward.  Counters run in their own passes with --kernel-trace only.
bash tools/pmc_halo.sh 31,512,256,256,128 > /dev/null 2>&1
bash tools/pmc_halo.sh 40,384,256,256,512 > /dev/null 2>&1
bash tools/pmc_wgrad_halo.sh pmc_wgrad_halo_r06 > gpurun_out/pmc_wgrad_halo_r06.txt 2>&1
PMC_CASES="mlp_shared_fwd_ps mlp_shared_fwd_merged3_ps mlp_shared_dgrad_merged_ps mlp_shared_wgrad_merged_ps gamma_beta_fwd_ps gamma_beta_dgrad_ps gamma_beta_wgrad_ps" bash tools/pmc_quick.sh pmc_r06 > gpurun_out/pmc_quick_r06.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r06 profiles/r05/pmc/traffic.json > gpurun_out/pmc_quick_r06.txt 2>&1
rm -rf gpurun_out/pmc_wgrad_halo_r06/tmp
tail -12 gpurun_out/pmc_quick_r06.txt; cat gpurun_out/pmc_halo_31_512_256_256_128.txt gpurun_out/pmc_halo_40_384_256_256_512.txt
