# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# rocprofv3 kernel statistics of 3 timed c3 steps (+1 warm-up) -> gpurun_out/prof_<tag>/, then the plain bench line.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fp32-leg > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
f=$(ls gpurun_out/prof_$TAG/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f gpurun_out/prof_$TAG/kernel_stats.csv && head -45 $f | cut -c1-200
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-leg > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -c 600 gpurun_out/bench_$TAG.json
