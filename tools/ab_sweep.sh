# usage (GPU box): bash tools/ab_sweep.sh "VAR1=val VAR2=val" ... ; each argument is one environment variant, "" = baseline.
# Prints ms/step of `bench.py --steps 4 --warmup 2` per variant, twice (boxes drift by ~1 % between runs).
for rep in 1 2; do
for v in "$@"; do
  r=$(env $v python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "rep$rep [$v] $r"
done
done
