"""Summarise a rocprofv3 kernel_stats.csv of tools/profile_round.sh by kernel family (per step).  usage: prof_summary.py <csv> [steps=4]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('GPU busy per step %.1f ms, dispatches/step %.0f' % (tot / steps / 1e6, sum(int(r['Calls']) for r in rows) / steps))
fam = {}
def family(n):
    if 'conv_' in n and 'layout' not in n: return 'conv'
    if 'norm' in n or 'bn_stats' in n or 'pair_sum' in n or 'channel_sum' in n: return 'norm'
    if any(k in n for k in ('gemm_f32', 'triple', 'embedding', 'col_sum', 'relu_bwd', 'graph_program')): return 'graph'
    if 'sn_' in n or 'dot_partial' in n: return 'spectral'
    if 'at::' in n or 'rocclr' in n: return 'aten'
    if any(k in n for k in ('pack_', 'transpose', 'prep')): return 'pack'
    if any(k in n for k in ('layout', 'pyramid', 'nearest', 'warp', 'avgpool', 'maxpool')): return 'layout/resample'
    return 'other'
for r in rows:
    k = family(r['Name'])
    d = fam.setdefault(k, [0.0, 0.0])
    d[0] += int(r['TotalDurationNs']) / steps / 1e6
    d[1] += int(r['Calls']) / steps
for k, (ms, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print('%-16s %8.1f ms %8.0f launches' % (k, ms, c))
