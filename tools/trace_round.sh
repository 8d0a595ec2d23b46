# usage (GPU box): bash tools/trace_round.sh <tag> -- rocprofv3 kernel trace (timestamps) of 1 warm-up + 2 timed c3 steps, then tools/trace_gaps.py
TAG=${1:-t}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$TAG -- python3 $R/bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fp32-leg > $R/gpurun_out/trace_$TAG.log 2>&1
cd $R
f=$(ls gpurun_out/trace_$TAG/*/*kernel_trace.csv | head -1)
head -1 $f | cut -c1-400
python tools/trace_gaps.py $f 0.34
