# usage (on the GPU box, from the repo root): bash tools/evidence_r06.sh <tag>
# Round 6 evidence: (a) rocprofv3 kernel statistics of 3 timed c3 steps at HEAD, (b) the default bench line,
# (c) the same line with the collectives live -- AG2V_FORCE_DIST=1 --gpus 1 --sync-bn 1: RCCL process group, synchronised batch norm on
# every fused path, the two-phase gradient all-reduce -- i.e. what ONE rank of an 8-GPU run executes, measured on the one GPU there is.
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fp32-leg > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
f=$(ls gpurun_out/prof_$TAG/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f gpurun_out/c3_step_kernel_stats_$TAG.csv && python tools/prof_summary.py $f > gpurun_out/prof_summary_$TAG.txt 2>&1
rm -rf gpurun_out/prof_$TAG
AG2V_TIMING_DETAIL=130 AG2V_TIMING_SCOPES=1 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_c3_default_$TAG.json 2> gpurun_out/c3_conv_by_shape_$TAG.txt
AG2V_FORCE_DIST=1 python bench.py --gpus 1 --sync-bn 1 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg > gpurun_out/bench_c3_forcedist_syncbn_$TAG.json 2> gpurun_out/bench_c3_forcedist_syncbn_$TAG.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg > gpurun_out/bench_c3_plain_after_$TAG.json 2>/dev/null
python - <<PY
import json
for n in ('bench_c3_default_$TAG', 'bench_c3_forcedist_syncbn_$TAG', 'bench_c3_plain_after_$TAG'):
    try:
        d = json.loads(open('gpurun_out/%s.json' % n).read().strip().splitlines()[-1])
        print(n, d['ms_per_step'], d['value'], d['roofline'].get('frac'), d['roofline'].get('spade_stack_frac'), d['comm'])
    except Exception as e:
        print(n, 'unreadable', e)
PY
