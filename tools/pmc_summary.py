"""Counter CSVs of tools/pmc_quick.sh -> <dir>/traffic.json, the file bench.py reads for `roofline.traffic`.
usage: pmc_summary.py <dir> [earlier traffic.json whose entries are kept for kernels this pass did not measure]
Units as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
64-byte requests as 32 bytes, so it is doubled.  One case per process (tools/pmc_traffic.py <case>): within a case the kernel name
identifies the launches of interest; FETCH and WRITE of a launch are joined on the dispatch id (the launch sequence is deterministic),
and the launches with the largest traffic (the case's own shape, not the helper launches around it) are averaged."""
import collections, csv, json, os, sys
d = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(path, counter):
    val, name = collections.OrderedDict(), {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            val[r['Dispatch_Id']] = val.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
            name[r['Dispatch_Id']] = r['Kernel_Name']
    return val, name


# (case, tag in the case's expectation file, kernel-name substring)
MATCH = [('mlp_shared_fwd', 'mlp_shared_fwd', 'conv_igemm_bf16x3_ws_kernel<128, 256'),
         ('mlp_shared_dgrad_merged', 'mlp_shared_dgrad_merged', 'conv_igemm_bf16x3_ws_kernel<256, 128'),
         ('mlp_shared_wgrad_merged', 'mlp_shared_wgrad_merged', 'conv_wgrad_bf16x3_v_kernel<128, 256'),
         ('conv_256_wgrad_128px', 'conv_256_wgrad_128px', 'conv_wgrad_bf16x3_v_kernel<256, 128'),
         ('mlp_shared_fwd_ps', 'mlp_shared_fwd_ps', 'conv_igemm_bf16x3_ps_kernel<128, 256'),
         # round 4: the same split-word launches run on the halo-tile LDS-DMA kernels (csrc/conv_halo.hip)
         ('mlp_shared_fwd_ps', 'mlp_shared_fwd_ps', 'conv_halo16_bf16x3_kernel<1, 8, 4, 128, 2, true>'),
         ('mlp_shared_dgrad_merged_ps', 'mlp_shared_dgrad_merged_ps', 'conv_halo16_bf16x3_kernel<2, 4, 4, 128, 1, true>'),
         ('mlp_shared_dgrad_merged_ps', 'mlp_shared_dgrad_merged_ps', 'conv_igemm_bf16x3_ps_kernel<256, 128'),
         ('mlp_shared_wgrad_merged_ps', 'mlp_shared_wgrad_merged_ps', 'conv_wgrad_bf16x3_vps_kernel<128, 256'),
         ('conv_256_wgrad_128px_ps', 'conv_256_wgrad_128px_ps', 'conv_wgrad_bf16x3_vps_kernel<256, 128'),
         # round 5: the LDS-DMA weight-gradient kernel (csrc/conv_wgrad_halo.hip), the exact-fp32 weight gradient, the gamma | beta layer
         ('mlp_shared_wgrad_merged_ps', 'mlp_shared_wgrad_merged_ps', 'conv_wgrad_halo_bf16x3_kernel<true, true>'),
         ('mlp_shared_wgrad_merged_fp32', 'mlp_shared_wgrad_merged_fp32', 'conv_wgrad_f32_v_kernel<128, 256'),
         ('gamma_beta_wgrad_ps', 'gamma_beta_wgrad_ps', 'conv_wgrad_halo_bf16x3_kernel<true, true>'),
         ('gamma_beta_dgrad_ps', 'gamma_beta_dgrad_ps', 'conv_halo16_bf16x3_kernel<1, 8, 4, 128, 2, true>'),
         ('gamma_beta_fwd_ps', 'gamma_beta_fwd_ps', 'conv_halo16_bf16x3_kernel<1, 8, 4, 128, 2, true>'),
         ('mlp_shared_dgrad_merged_ps', 'mlp_shared_dgrad_merged_ps', 'conv_halo16_bf16x3_kernel<1, 8, 4, 128, 2, true>'),
         # round 6: a block's three mlp_shared convolutions as one forward launch (ag2v_conv_opts::out_mstride)
         ('mlp_shared_fwd_merged3_ps', 'mlp_shared_fwd_merged3_ps', 'conv_halo16_bf16x3_kernel<1, 8, 4, 128, 2, true>'),
         ('layout_conv', 'layout_conv_fwd', 'layout_conv_fwd3_kernel<16>'),
         ('flow_res_h16', 'flow_res_h16', 'conv_halo16_f16x3_kernel<1, 8, 4, 32, 8>'),
         ('flow_res_h16', 'flow_res_h16', 'conv_igemm_f16x3_ws_kernel<256, 128'),
         ('bn_pyramid', 'bn_pyr_bwd_reduce', 'norm_bwd_reduce_kernel<0, true, true>'), ('bn_pyramid', 'bn_pyr_bwd_apply', 'norm_bwd_apply_kernel<0, true, true>'),
         ('layout_conv_bwd', 'layout_conv_bwd', 'layout_conv_bwd_kernel<3>'),
         ('layout_conv', 'layout_conv_fwd', 'layout_conv_fwd_kernel<3, 16>'), ('flow_res_fp32', 'flow_res_fp32', 'conv_igemm_f32_ws_kernel<256, 128'),
         ('norm', 'norm_apply', 'norm_apply_kernel<1, true>'), ('norm', 'norm_bwd_reduce', 'norm_bwd_reduce_kernel<1, true>'),
         ('norm', 'norm_bwd_apply', 'norm_bwd_apply_kernel<1, true>'), ('norm', 'bn_stats_partial', 'bn_stats_partial_kernel<true>'),
         ('spade_pair', 'spade_pair_apply', 'spade_pair_apply_kernel<2>'), ('spade_pair', 'spade_pair_bwd_reduce', 'spade_pair_bwd_reduce_kernel<2>'),
         ('spade_pair', 'spade_pair_bwd_apply', 'spade_pair_bwd_apply_kernel<2>')]
out = {}
for case, tag, sub in MATCH:
    fp, wp = os.path.join(d, '%s_FETCH_SIZE_counter_collection.csv' % case), os.path.join(d, '%s_WRITE_SIZE_counter_collection.csv' % case)
    ep = os.path.join(root, 'gpurun_out', 'pmc_expect_%s.json' % case)
    if not (os.path.exists(fp) and os.path.exists(wp) and os.path.exists(ep)):
        continue
    expect = json.load(open(ep))[tag]
    f, fn = per_dispatch(fp, 'FETCH_SIZE')
    w, wn = per_dispatch(wp, 'WRITE_SIZE')
    ids = [i for i in f if sub in fn[i] and i in w and sub in wn[i]]
    if not ids:
        continue
    tot = {i: 2 * 1024 * f[i] + 1024 * w[i] for i in ids}
    top = max(tot.values())
    ids = [i for i in ids if tot[i] >= 0.8 * top]
    key = fn[ids[0]].split('(')[0].replace('void ', '')
    if key in out and out[key]['shape'] != expect['shape']:
        key = '%s @ %s' % (key, case)            # the same kernel on another layer shape: its own entry (bench.py reads the plain key)
    fb, wb = 2 * 1024 * sum(f[i] for i in ids) / len(ids), 1024 * sum(w[i] for i in ids) / len(ids)
    out[key] = {'shape': expect['shape'], 'fetch_bytes': fb, 'write_bytes': wb, 'algorithmic_bytes': expect['algorithmic_bytes'],
                'launches': len(ids), 'ratio': (fb + wb) / expect['algorithmic_bytes']}
    print('%-72s FETCH %8.0f MB  WRITE %8.0f MB  algorithmic %8.0f MB  x%.2f   [%s]' % (
        key[:72], fb / 1e6, wb / 1e6, expect['algorithmic_bytes'] / 1e6, out[key]['ratio'], expect['shape']))
# kernels that were not re-measured in this pass keep the newest earlier round's entry (marked so): bench.py reads ONE file
base = None
if len(sys.argv) > 2:
    base = json.load(open(sys.argv[2]))
    for k, v in base.items():
        if k not in out:
            v = dict(v)
            v.setdefault('measured_in', os.path.relpath(sys.argv[2], root))
            out[k] = v
json.dump(out, open(os.path.join(d, 'traffic.json'), 'w'), indent=1)
