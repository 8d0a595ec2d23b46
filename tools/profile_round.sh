cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v29 -- python3 $R/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fp32-leg > $R/gpurun_out/prof_v29.log 2>&1
cd $R
python bench.py > gpurun_out/bench_default_v29.json 2> gpurun_out/bench_default_v29.err
tail -c 1500 gpurun_out/bench_default_v29.json
