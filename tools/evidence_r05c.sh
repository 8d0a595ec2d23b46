# The commands behind profiles/r05/*_r05c.* (end of round 5; GPU box, repo root): bash tools/evidence_r05c.sh
set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_c3_default_r05c.json 2> gpurun_out/bench_c3_default_r05c.err
for c in c1 c2 c5; do python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg > gpurun_out/bench_${c}_r05c.json 2>/dev/null; done
AG2V_TIMING_DETAIL=400 AG2V_TIMING_SCOPES=1 python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-fp32-leg 2> gpurun_out/c3_conv_by_shape_r05c.txt > /dev/null
bash tools/profile_round.sh r05d > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/prof_r05d/kernel_stats.csv 4 > gpurun_out/prof_summary_r05c.txt
PMC_CASES="layout_conv bn_pyramid norm" bash tools/pmc_quick.sh pmc_r05c > gpurun_out/pmc_quick_r05c.txt 2>&1
tail -c 400 gpurun_out/bench_c3_default_r05c.json
