set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_c3_default_r04e.json 2> gpurun_out/bench_c3_default_r04e.err
for c in c1 c2 c5; do python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg > gpurun_out/bench_${c}_r04e.json 2>/dev/null; done
AG2V_TIMING_DETAIL=400 AG2V_TIMING_SCOPES=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fp32-leg 2> gpurun_out/c3_conv_by_shape_r04e.txt > /dev/null
bash tools/pmc_lc.sh fwd3 > /dev/null 2>&1
AG2V_NO_LC_FWD3=1 bash tools/pmc_lc.sh second_form > /dev/null 2>&1
tools/ubench/lc_stamps > gpurun_out/lc_stamps_second_form.txt 2>&1; tools/ubench/lc_stamps img >> gpurun_out/lc_stamps_second_form.txt 2>&1
tools/ubench/store_pattern > gpurun_out/store_pattern.txt 2>&1; python tools/hbm_write_probe.py >> gpurun_out/store_pattern.txt 2>&1
python tools/lc_bench.py > gpurun_out/lc_bench_fwd3.txt 2>&1; AG2V_NO_LC_FWD3=1 python tools/lc_bench.py > gpurun_out/lc_bench_second_form.txt 2>&1
python tools/fewrow_wgrad_bench.py > gpurun_out/fewrow_wgrad_bench.txt 2>&1
tail -c 400 gpurun_out/bench_c3_default_r04e.json
