"""FETCH_SIZE / WRITE_SIZE counter CSVs of tools/pmc_quick.sh -> <dir>/traffic.json, the file bench.py reads for `roofline.traffic`.
Units as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes: both counters are in KiB; on gfx950 FETCH_SIZE counts 64-byte
requests as 32 bytes, so it is doubled.  Per kernel: the launches whose traffic is largest (the big shapes of tools/pmc_traffic.py;
warm-up / helper launches of the same kernel on other shapes are ignored) averaged."""
import collections, csv, json, os, sys
d = sys.argv[1]
expect = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'pmc_expect.json')))


def per_dispatch(path, counter):
    val, name = collections.defaultdict(float), {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            val[r['Dispatch_Id']] += float(r['Counter_Value'])
            name[r['Dispatch_Id']] = r['Kernel_Name']
    per = collections.defaultdict(list)
    for k, v in val.items():
        per[name[k]].append(v)
    return per


f = per_dispatch(os.path.join(d, 'fetch_size_counter_collection.csv'), 'FETCH_SIZE')
w = per_dispatch(os.path.join(d, 'write_size_counter_collection.csv'), 'WRITE_SIZE')
# which expectation belongs to which kernel (substring of the kernel name, pass)
MATCH = [('mlp_shared_fwd', 'conv_igemm_bf16x3_ws_kernel<128, 256, 2, 4, 8, false, true, true>'),
         ('mlp_shared_dgrad_merged', 'conv_igemm_bf16x3_ws_kernel<256, 128, 4, 2, 8, false, true, false>'),
         ('mlp_shared_wgrad_merged', 'conv_wgrad_bf16x3_v_kernel<256, 128, 4, 2, 8, false>'),
         ('gamma_beta_wgrad_256', 'conv_wgrad_bf16x3_v_kernel<128, 256, 2, 4, 8, false>'),
         ('norm_apply', 'norm_apply_kernel<1, true>'), ('norm_bwd_reduce', 'norm_bwd_reduce_kernel<1, true>'),
         ('norm_bwd_apply', 'norm_bwd_apply_kernel<1, true>'), ('bn_stats_partial', 'bn_stats_partial_kernel<true>')]
out = {}
for tag, sub in MATCH:
    names = [k for k in f if sub in k]
    if not names:
        continue
    k = names[0]
    fv, wv = sorted(f[k], reverse=True), sorted(w.get(k, [0.0]), reverse=True)
    top_f = [x for x in fv if x >= 0.8 * fv[0]]
    top_w = [x for x in wv if x >= 0.8 * wv[0]] or [0.0]
    key = k.split('(')[0].replace('void ', '')
    out[key] = {'shape': expect[tag]['shape'], 'fetch_bytes': 2 * 1024 * sum(top_f) / len(top_f), 'write_bytes': 1024 * sum(top_w) / len(top_w),
                'algorithmic_bytes': expect[tag]['algorithmic_bytes'], 'launches': len(top_f),
                'ratio': (2 * 1024 * sum(top_f) / len(top_f) + 1024 * sum(top_w) / len(top_w)) / expect[tag]['algorithmic_bytes']}
    print('%-70s FETCH %8.0f MB  WRITE %8.0f MB  algorithmic %8.0f MB  x%.2f   [%s]' % (
        key[:70], out[key]['fetch_bytes'] / 1e6, out[key]['write_bytes'] / 1e6, out[key]['algorithmic_bytes'] / 1e6, out[key]['ratio'], expect[tag]['shape']))
json.dump(out, open(os.path.join(d, 'traffic.json'), 'w'), indent=1)
