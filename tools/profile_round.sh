cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v20 -- python3 $R/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fp32-leg > $R/gpurun_out/prof_v20.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_v20_fetch -- python3 $R/tools/pmc_traffic.py > $R/gpurun_out/pmc_v20_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_v20_write -- python3 $R/tools/pmc_traffic.py > $R/gpurun_out/pmc_v20_write.log 2>&1
cd $R
python bench.py > gpurun_out/bench_default_v20.json 2> gpurun_out/bench_default_v20.err
tail -c 1500 gpurun_out/bench_default_v20.json
