# The commands behind profiles/r05/* (GPU box, repo root): bash tools/evidence_r05.sh
set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_c3_default_r05.json 2> gpurun_out/bench_c3_default_r05.err
for c in c1 c2 c5; do python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg > gpurun_out/bench_${c}_r05.json 2>/dev/null; done
AG2V_TIMING_DETAIL=400 AG2V_TIMING_SCOPES=1 python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-fp32-leg 2> gpurun_out/c3_conv_by_shape_r05.txt > /dev/null
bash tools/profile_round.sh r05 > /dev/null 2>&1
python tools/wgrad_halo_bench.py 20 > gpurun_out/wgrad_halo_bench_r05.txt 2>&1
WGH_CUSTOM="120,64,256,256,128,15,1;120,256,128,128,128,15,1;120,1024,32,32,512,15,1;120,64,256,256,128,1,2;120,512,64,64,256,1,2;120,256,32,32,256,15,0;120,64,128,128,128,15,0;120,64,256,256,64,15,1" python tools/wgrad_halo_bench.py 20 >> gpurun_out/wgrad_halo_bench_r05.txt 2>&1
bash tools/pmc_wgrad_halo.sh pmc_wgrad_halo_r05 > gpurun_out/pmc_wgrad_halo_r05.txt 2>&1
WGH_CUSTOM="2,512,256,256,384;21,512,256,256,384;60,128,256,256,256" bash tools/pmc_fetch_wgrad.sh pmc_fetch_wgrad_r05 > gpurun_out/pmc_fetch_wgrad_r05.txt 2>&1
bash tools/pmc_quick.sh pmc_r05 > gpurun_out/pmc_quick_r05.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r05 profiles/r04/pmc/traffic.json >> gpurun_out/pmc_quick_r05.txt 2>&1
tail -c 400 gpurun_out/bench_c3_default_r05.json
