#!/usr/bin/env python
"""Benchmark of the AG2Video training step (generator + discriminator forward/backward + Adam) on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1            # c3: CATER 256x256, 16 frames, batch 8 per GPU
    python bench.py --gpus N --steps K --warmup W            # starts the N ranks itself (a child torch.distributed.run), relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W               # one rank per GPU over RCCL, weak scaling

A "step" is one pass of scripts/train.py:446-464,522-525 (G fwd, G loss incl. two D passes, G bwd, Adam(G), D loss with two
D passes, D bwd, Adam(D)) over one synthetic CATER-shaped clip batch that is already resident in HBM.  Rank 0 prints ONE
JSON line: frames/sec/node plus the roofline figures of the dominant kernel and a CPU baseline (the oracle's step timed
on the host cores at a bounded size).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402

CONFIGS = {
    # name: (image side, frames, per-GPU batch, extra flags)
    'c1': (64, 4, 1, []),
    'c2': (128, 8, 4, []),
    'c3': (256, 16, 8, []),
    # c5: Something-Something-V2 / Something-Else shaped (data/smth.py:85-104): ONE attribute => 128 layout channels, 2-3
    # objects per clip incl. the hand, 9 predicates, 9 action classes, 224 x 224 (= 7 * 32), 16 frames
    'c5': (224, 16, 8, ['--dataset', 'smth_else']),
    'toy': (64, 3, 2, ['--ngf', '8', '--ndf', '8', '--nff', '8', '--gconv_hidden_dim', '64']),
}
PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide: dense bf16 MFMA peak; the bf16x3 path issues 3 MFMAs per product => /3
# HBM-side traffic of the conv / norm kernels: rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of
# tools/pmc_traffic.py through tools/pmc_quick.sh; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md), summarised
# by tools/pmc_summary.py into profiles/<round>/pmc/traffic.json next to the counter CSVs it was computed from.  bench.py READS that
# file (newest round first): {kernel name: {"shape": ..., "fetch_bytes", "write_bytes", "algorithmic_bytes", "launches"}} -- bytes
# per launch ON THE SHAPE NAMED there, not an average over the shapes of a training step.
def load_pmc_traffic():
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*', 'pmc', 'traffic.json')), reverse=True):
        try:
            return json.load(open(path)), os.path.relpath(path, REPO)
        except Exception:
            continue
    return {}, None


PEAK_HBM_GBS = 8000.0              # same guide: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def make_opt(cfg, vgg=False):
    from ag2video_amd.data.args import make_opt as _mk
    from ag2video_amd.data.vocab import cater_vocab, smth_vocab
    side, frames, batch, extra = CONFIGS[cfg]
    # vgg: the perceptual loss on a RANDOM-INIT VGG19 (--vgg_random_init: the ImageNet weights are not obtainable here) -- a memory / time
    # probe of the full loss set, not BASELINE.json's metric (which is quoted with the VGG loss off)
    flags = (['--vgg_random_init'] if vgg else ['--no_vgg_loss']) + ['--image_size', '%d,%d' % (side, side), '--batch_size', str(batch), '--gpu_ids', '0'] + extra
    vocab = smth_vocab() if 'smth_else' in extra else cater_vocab()
    return _mk(vocab, flags), side, frames, batch


def make_clip(opt, cfg, batch, frames, side, seed):
    """Seeded synthetic clip batch of the configuration's dataset shape: CATER 5-10 objects / <= 6 actions;
    Something-Else 2-3 objects (the hand last) / <= 2 actions."""
    from ag2video_amd.data.synthetic import make_clip_batch
    if getattr(opt, 'dataset', 'cater') == 'smth_else':
        return make_clip_batch(opt.vocab, batch, frames, (side, side), seed=seed, min_objs=2, max_objs=3, max_actions=2)
    return make_clip_batch(opt.vocab, batch, frames, (side, side), seed=seed)


def cpu_baseline_child():
    """Runs in a child process without a GPU: the oracle (CPU restatement of the reference step) timed on this host's cores on two
    bounded samples (SURVEY.md 8d "CPU baseline"): c1 (64x64, 4 frames, batch 1: the reference's own CPU-runnable configuration) and
    c3' = the GPU workload's resolution at the smallest clip that still has its structure (256x256, 4 frames, batch 1; full c3 is
    ~1 h per step on a CPU).  Default widths, full G + D step each."""
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.models.meta_models import AG2VideoModel, MetaDiscriminatorModel
    from oracle.step import OracleTrainer
    threads = max(1, min(16, os.cpu_count() or 1))   # the step is thousands of small ops: more threads only add barriers
    torch.set_num_threads(threads)
    cpu = torch.device('cpu')
    t_start = time.time()

    def leg(side, frames, batch, warm, max_steps, budget_s):
        opt, _, _, _ = make_opt('c1')
        opt.image_size = (side, side)
        opt.batch_size = batch
        opt.gpu_ids = []
        shapes_G = {k: list(v.shape) for k, v in AG2VideoModel(opt, cpu).state_dict().items()}
        shapes_D = {k: list(v.shape) for k, v in MetaDiscriminatorModel(opt).img_discriminator.state_dict().items()}
        tr = OracleTrainer(opt, shapes_G, shapes_D, 0)
        b = make_clip_batch(opt.vocab, batch, frames, (side, side), seed=1234)
        w = 0.0
        if warm:        # one untimed step (first-touch allocations, primitive caches: 2.8 s on one box and 22 s on another at c1)
            t0 = time.time()
            tr.gd_step(b)
            w = time.time() - t0
        n, t0 = 0, time.time()
        while n < max_steps and (n == 0 or time.time() - t0 < budget_s):
            tr.gd_step(b)
            n += 1
        dt = (time.time() - t0) / n
        return {'frames_per_sec': round(batch * frames / dt, 4), 's_per_step': round(dt, 2), 'steps': n, 'warm_s': round(w, 1)}
    c1 = leg(64, 4, 1, True, 3, 10.0)
    res = {"value": c1['frames_per_sec'], "unit": "frames/sec", "cores": threads, "kind": "port", "c1": c1, "c3prime": None,
           "sample": "oracle (PyTorch-CPU restatement of the reference), full steps (G fwd+loss+bwd+Adam, D loss+bwd+Adam), default "
                     "widths: c1 (64x64, 4 frames, batch 1) %d steps of %.1f s after one untimed step of %.1f s"
                     % (c1['steps'], c1['s_per_step'], c1['warm_s']),
           #  how this port relates to the reference's own modules: BASELINE.md section 2 timed the IMPORTED
           # reference in the build container -- 8 cores, torch 2.10 CPU, first-call oneDNN primitive creation inside the window
           "note": "kind 'port' = this repo's CPU restatement (oracle/), pinned to the reference by golden vectors; the imported reference itself "
                   "measured 0.050 frames/s at c3' and 0.20 at c1 on 8 cores in the build container (BASELINE.md section 2): the port on this "
                   "host's cores, warmed (one untimed step), is ~4-5x that -- same arithmetic, twice the cores, no first-call primitive creation "
                   "in the window, no Python loops over samples"}
    print(json.dumps(res), flush=True)                       # a complete line already: the c3' leg may hit the parent's time limit
    if os.environ.get('AG2V_CPU_BASELINE_C3P', '1') != '0' and time.time() - t_start < 60:
        c3p = leg(256, 4, 1, True, 2, 25.0)                  # one untimed step, then 2 timed ones (a second only when the first took < 25 s)
        res['c3prime'] = c3p
        res['value'] = c3p['frames_per_sec']                 # the leg at the GPU workload's resolution is the headline value
        res['sample'] += "; c3' (256x256, 4 frames, batch 1: the GPU workload's resolution) %d step(s) of %.1f s on average after one untimed step of %.1f s = %.4f frames/s -- this is `value`; c1: %.4f frames/s" % (
            c3p['steps'], c3p['s_per_step'], c3p['warm_s'], c3p['frames_per_sec'], c1['frames_per_sec'])
        print(json.dumps(res), flush=True)


def cpu_baseline(timeout_s=360):
    """Spawn the CPU baseline as a child process (bounded wall time; the GPU process itself never runs the oracle)."""
    import subprocess
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='', OMP_NUM_THREADS='16', MKL_NUM_THREADS='16')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only'], env=env, capture_output=True,
                             text=True, timeout=timeout_s)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        return {"value": None, "unit": "frames/sec", "cores": None, "kind": "port",
                "sample": "cpu baseline child failed: %s" % (out.stderr.strip()[-200:],)}
    except subprocess.TimeoutExpired as e:
        txt = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or '')
        for line in reversed(txt.strip().splitlines()):
            if line.startswith('{'):
                d = json.loads(line)
                d['sample'] += " (the c3' leg exceeded the %d s limit and was stopped)" % timeout_s
                return d
        return {"value": None, "unit": "frames/sec", "cores": None, "kind": "port",
                "sample": "cpu baseline child exceeded %d s and was stopped" % timeout_s}


def launch_ranks(n):
    """`python bench.py --gpus N` run directly (no torchrun environment): start N ranks, one per GPU, as a CHILD `python -m
    torch.distributed.run` of this same file and relay rank 0's JSON line.  The reference's counterpart is `--gpu_ids 0,..,N-1`
    -> DataParallelWithCallback(device_ids=...) inside one process (models/meta_models.py:16-27, scripts/train.py:375); here it is
    one process per GPU over RCCL.  The parent never initialises the GPU (device_count() does not, on this image) and never
    exec()s: the children are ordinary subprocesses and their exit code is ours."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        print("bench.py: --gpus %d requested but %d GPU(s) are visible; refusing to run fewer ranks than asked for" % (n, have), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)          # stderr passes straight through
    lines = []
    for line in proc.stdout:
        if line.startswith('{'):
            lines.append(line.rstrip('\n'))
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc != 0:
        print("bench.py: the %d-rank run failed (torch.distributed.run exit code %d)" % (n, rc), file=sys.stderr)
        return rc if 0 < rc < 256 else 1
    if len(lines) != 1:
        print("bench.py: expected ONE JSON line from rank 0, got %d" % len(lines), file=sys.stderr)
        return 1
    d = json.loads(lines[0])
    if not (d.get('n_gpus') == n and d.get('comm', {}).get('rccl_world') == n):
        print("bench.py: asked for %d ranks, the line reports n_gpus=%r rccl_world=%r" % (n, d.get('n_gpus'), d.get('comm', {}).get('rccl_world')), file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)     # warm-up 3 steps, time >= 10 steps (~20 s at c3)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='c3', choices=sorted(CONFIGS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--precision', default='bf16x3', choices=['bf16x3', 'fp32'],
                    help='arithmetic of the 128-row conv MFMA tiles (both hold the 1e-3 parity bound; tests/test_step_gpu.py)')
    ap.add_argument('--no-fp32-leg', action='store_true', help='skip the extra exact-fp32 step used for roofline_exact_fp32')
    ap.add_argument('--temporal-D', action='store_true', help='--temporal_D 1: the temporal multiscale PatchGAN on top of the image discriminator (not BASELINE.json\'s metric; reported in config.workload)')
    ap.add_argument('--vgg-random-init', action='store_true', help='perceptual loss on a random-init VGG19 (memory / time probe; reported in config.workload)')
    ap.add_argument('--sync-bn', type=int, default=-1, choices=[-1, 0, 1],
                    help="batch-norm statistics over all ranks with the reference's multi-device formula: -1 (default) = as the reference at this "
                         "world size (on at N > 1, off at N = 1); the N > 1 line then also carries the per-rank-statistics step time")
    ap.add_argument('--lean-tape', action='store_true', help='--lean_tape 1: smaller autograd tape for a few extra passes over memory (reported in config.workload)')
    args = ap.parse_args()
    if args.cpu_baseline_only:
        cpu_baseline_child()
        return
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if 'RANK' not in os.environ and (args.gpus > 1 or os.environ.get('AG2V_FORCE_DIST') == '1'):
        sys.exit(launch_ranks(args.gpus))       # this process has not touched the GPU: it only starts the ranks and relays rank 0's line
    env_world = int(os.environ.get('WORLD_SIZE', '1'))
    if env_world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, env_world))

    from ag2video_amd import distributed as D
    from ag2video_amd import ops
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.engine import TrainStep
    rank, world, local = D.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    opt, side, frames, batch = make_opt(args.config, vgg=args.vgg_random_init)
    opt.gpu_ids = [local]
    opt.temporal_D = 1 if args.temporal_D else 0
    opt.lean_tape = 1 if args.lean_tape else 0
    opt.sync_bn = args.sync_bn
    ops.set_conv_precision(args.precision)
    ts = TrainStep(opt, device, weights_seed=0)
    clip = ts.to_device(make_clip(opt, args.config, batch, frames, side, seed=1234 + rank))
    torch.cuda.synchronize()

    for i in range(args.warmup):
        if not args.no_kernel_timing and i == args.warmup - 1:
            ops.enable_kernel_timing(True)          # the last warm-up step also creates the HIP events the timed steps reuse
        ts.gd_step(clip)
    torch.cuda.synchronize()
    if D.is_dist():
        torch.distributed.barrier()
    if not args.no_kernel_timing:
        if args.warmup > 0:
            ops.fold_kernel_timing(keep=False)      # discard the warm-up records, keep their events
        else:
            ops.enable_kernel_timing(True)
    torch.cuda.synchronize()
    if D.is_dist():
        ts.comm_events = []                     # event pairs around the gradient all-reduce joins (engine.gd_step)
    t0 = time.time()
    for i in range(args.steps):
        # HIP event pairs around every convolution call cost ~3 % of a step (20k stream markers): they are recorded during the
        # LAST timed step only, which is enough for the per-kernel table (~10k launches)
        ops.pause_kernel_timing(args.no_kernel_timing or i + 1 < args.steps)
        model_out, GL, DL = ts.gd_step(clip)
    torch.cuda.synchronize()
    if D.is_dist():
        torch.distributed.barrier()
    dt = D.max_over_ranks(time.time() - t0, device)
    ts.check_health()                               # (after the clock stopped) the last step's graph-net programs completed their barriers
    timing = ops.kernel_timing_summary() if not args.no_kernel_timing else {}
    by_shape = ops.kernel_timing_summary(by_shape=True) if not args.no_kernel_timing else {}
    by_scope = ops.kernel_timing_summary(by_scope=True) if not args.no_kernel_timing else {}
    # ---- communication record (N > 1): proves how many ranks RCCL saw and how much of the all-reduce time the step waits for
    comm = {"rccl_world": (torch.distributed.get_world_size() if D.is_dist() else 1),
            "backend": (torch.distributed.get_backend() if D.is_dist() else None)}
    if D.is_dist():
        ev, ts.comm_events = ts.comm_events, None
        per = {}
        for name, e0, e1 in ev:
            per.setdefault(name, []).append(e0.elapsed_time(e1))
        gflat, dflat = ts.optimizer_generator.flat_grad, ts.discriminator.optimizer_d_img.flat_grad

        def sync_ms(flat):          # the same buckets, NOT overlapped with anything: one untimed call between two events
            torch.cuda.synchronize()
            torch.distributed.barrier()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record()
            D.average_gradients([], flat=flat)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b)
        g_alone, d_alone = sync_ms(gflat), sync_ms(dflat)
        g_exposed = sum(per.get('g_join', [0.0])) / max(1, len(per.get('g_join', [])))
        comm.update({
            "g_allreduce_bytes": gflat.numel() * 4, "d_allreduce_bytes": dflat.numel() * 4,
            "g_allreduce_ms_alone": round(g_alone, 3), "d_allreduce_ms_alone": round(d_alone, 3),
            "g_allreduce_exposed_ms_per_step": round(g_exposed, 3),
            "g_allreduce_hidden_ms_per_step": round(max(0.0, g_alone - g_exposed), 3),
            "d_allreduce_exposed_ms_per_step": round(sum(per.get('d_sync', [0.0])) / max(1, len(per.get('d_sync', []))), 3),
            "g_bus_gbps_alone": round(2.0 * (world - 1) / world * gflat.numel() * 4 / (g_alone * 1e-3) / 1e9, 1) if world > 1 else None,
            "note": "exposed = time the compute stream waits at the join of the generator's asynchronous all-reduce (launched after G's "
                    "backward, joined behind the discriminator phase); alone = the same buckets reduced with nothing else running; the "
                    "discriminator's all-reduce is synchronous (fully exposed)"})
    comm["sync_bn"] = bool(D.sync_bn())
    if D.is_dist() and D.sync_bn():          # (world 1 under AG2V_FORCE_DIST=1 --sync-bn 1 as well: the one-GPU rehearsal of an 8-GPU rank)
        # the same step with PER-RANK batch statistics (F.batch_norm's formula; --sync-bn 0), beside the synchronised default
        D.set_sync_bn(False)
        ops.pause_kernel_timing(True)
        ts.gd_step(clip)
        torch.cuda.synchronize()
        torch.distributed.barrier()
        t1 = time.time()
        k2 = max(2, args.steps // 2)
        for i in range(k2):
            ts.gd_step(clip)
        torch.cuda.synchronize()
        torch.distributed.barrier()
        comm["per_rank_bn_ms_per_step"] = round(D.max_over_ranks(time.time() - t1, device) * 1e3 / k2, 2)
        comm["per_rank_bn_steps"] = k2
        D.set_sync_bn(True)
    if os.environ.get('AG2V_TIMING_DETAIL') and rank == 0:
        det = ops.kernel_timing_summary(by_shape=True, by_scope=os.environ.get('AG2V_TIMING_SCOPES') is not None)
        top = int(os.environ['AG2V_TIMING_DETAIL']) if os.environ['AG2V_TIMING_DETAIL'].isdigit() and int(os.environ['AG2V_TIMING_DETAIL']) > 1 else 45
        for k, v in sorted(det.items(), key=lambda kv: -kv[1]['ms'])[:top]:
            print('DETAIL %8.1f ms %5d launches %6.1f TF  %s' % (v['ms'], v['launches'], v['flops'] / (v['ms'] * 1e-3) / 1e12, k), file=sys.stderr)
    ops.enable_kernel_timing(False)
    finite = all(bool(torch.isfinite(v).all()) for v in list(GL.values()) + list(DL.values()))

    # graph-only step (train.py:471-493), reported separately
    gclip = ts.to_device(make_clip(opt, args.config, batch, frames, side, seed=4321 + rank))
    ts.graph_step(gclip)
    torch.cuda.synchronize()
    tg = time.time()
    ts.graph_step(gclip)
    torch.cuda.synchronize()
    graph_ms = (time.time() - tg) * 1e3

    fp32_leg = None
    if args.precision == 'bf16x3' and not args.no_fp32_leg and not args.no_kernel_timing and world == 1:
        ops.set_conv_precision('fp32')
        ts.gd_step(clip)                       # warm-up in the exact mode
        torch.cuda.synchronize()
        ops.enable_kernel_timing(True)
        t1 = time.time()
        ts.gd_step(clip)
        torch.cuda.synchronize()
        fp32_leg = {'ms': (time.time() - t1) * 1e3, 'timing': ops.kernel_timing_summary(), 'by_scope': ops.kernel_timing_summary(by_scope=True)}
        ops.enable_kernel_timing(False)
        ops.set_conv_precision(args.precision)

    if rank != 0:
        D.shutdown()
        return
    ms_per_step = dt / args.steps * 1e3
    value = world * batch * frames * args.steps / dt
    PMC, PMC_PATH = load_pmc_traffic()

    def top_shapes(kernel, n=3):
        """The dominant kernel's n shapes with the most time: (N, C, H, W, K, R, stride) of the call, its passes, TFLOP/s."""
        rows = {}
        for k, v in by_shape.items():
            if not k.startswith('conv_') or ':' not in k:
                continue
            lab, rest = k.split(':', 1)
            name, _, shp = rest.partition(' (')
            if name != kernel:
                continue
            d = rows.setdefault('(' + shp, {'ms': 0.0, 'flops': 0.0, 'launches': 0, 'passes': set()})
            d['ms'] += v['ms']; d['flops'] += v['flops']; d['launches'] += v['launches']; d['passes'].add(lab[5:])
        out = []
        for shp, d in sorted(rows.items(), key=lambda kv: -kv[1]['ms'])[:n]:
            tf = d['flops'] / (d['ms'] * 1e-3) / 1e12
            out.append({"shape_N_C_H_W_K_R_stride": shp, "passes": sorted(d['passes']), "launches": d['launches'], "ms": round(d['ms'], 2),
                        "tflops": round(tf, 1), "frac": round(tf / ((PEAK_BF16_MFMA_TFLOPS / 3.0) if 'bf16x3' in kernel else PEAK_FP32_MFMA_TFLOPS), 4)})
        return out

    def conv_roofline(tm, wall_ms):
        """Roofline block of the dominant conv kernel of a timing summary (work slot = dense-conv algorithmic FLOPs).
        Timing records are keyed "conv_<pass>:<HIP kernel name>"; forward and data gradient share the implicit-GEMM kernels, so
        the table is aggregated per HIP kernel -- the names and launch counts then line up with `rocprofv3 --kernel-trace --stats`
        (profiles/); the event pair of a record also spans the weight-prep / memset launches of that call (a few microseconds)."""
        per = {}
        for k, v in tm.items():
            if not k.startswith('conv_'):
                continue
            name = k.split(':', 1)[1] if ':' in k else k
            d = per.setdefault(name, {'ms': 0.0, 'flops': 0.0, 'launches': 0, 'passes': {}})
            d['ms'] += v['ms']; d['flops'] += v['flops']; d['launches'] += v['launches']
            d['passes'][k.split(':', 1)[0]] = {'ms': round(v['ms'], 2), 'launches': v['launches'],
                                               'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2)}
        if not per:
            return None
        dom = max(per, key=lambda k: per[k]['ms'])
        t = per[dom]
        split = 'bf16x3' in dom
        peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if split else PEAK_FP32_MFMA_TFLOPS
        achieved = t['flops'] / (t['ms'] * 1e-3) / 1e12
        conv_ms = sum(v['ms'] for v in per.values())
        conv_fl = sum(v['flops'] for v in per.values())
        pmc = PMC.get(dom)
        return {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4),
                "traffic": ((pmc['fetch_bytes'] + pmc['write_bytes']) if pmc else None),
                "traffic_note": (("%s: FETCH %.0f MB + WRITE %.0f MB per launch on shape %s vs %.0f MB algorithmic (x%.2f)"
                                  % (PMC_PATH, pmc['fetch_bytes'] / 1e6, pmc['write_bytes'] / 1e6, pmc['shape'],
                                     pmc['algorithmic_bytes'] / 1e6,
                                     (pmc['fetch_bytes'] + pmc['write_bytes']) / pmc['algorithmic_bytes'])) if pmc
                                 else 'no PMC pass for this kernel under profiles/'),
                "peak_basis": ("bf16 dense MFMA 2500 TF (2.4 GHz) / 3 MFMAs per fp32-equivalent product; under this instruction mix the chip holds "
                               "1.81 GHz on the LDS-DMA kernel (PMC, profiles/r06/pmc_halo_*_r06h.txt: matrix pipe 74 % busy), i.e. 628 TF "
                               "MFMA-bound at the held clock"
                               if split else "v_mfma_f32_32x32x2_f32 dense peak"),
                # a CALL = one ag2v_conv2d_*_ex entry; the library cuts a call whose operands exceed 4 GiB into sample chunks, so rocprofv3
                # counts more kernel LAUNCHES of this kernel than calls (c3: 25 calls = 38 launches per step).  achieved =
                # FLOPs per call / time per call = FLOPs per launch / time per launch either way.
                "calls": t['launches'], "avg_call_ms": round(t['ms'] / t['launches'], 4),
                "calls_cover": "one step (the last timed one)",
                "flops_per_call": t['flops'] / t['launches'], "passes": t['passes'],
                "by_shape": (top_shapes(dom) if tm is timing else None),
                "all_conv_kernels": {"achieved": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                                     "share_of_step_time": round(conv_ms / wall_ms, 4),
                                     "by_kernel": {k: {"ms": round(v['ms'], 2), "tflops": round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2),
                                                       "launches": v['launches']} for k, v in sorted(per.items())}}}

    def stack_roofline(scoped, peak, peak_basis):
        """north_star's own figure: MFMA-roofline fraction of the SPADE decoder conv stack = the convolution launches made
        inside SPADEGenerator.forward and their backward launches (netG: fc, every SPADEResnetBlock's conv_0 / conv_1 / conv_s and its
        SPADEs' mlp_shared / gamma / beta, conv_img; reference spade_generator.py:16-34, architecture.py:29-32, normalization.py:90-94).
        `algorithmic` = dense-conv 2 * MAC FLOPs of those launches (what the timing records carry; SURVEY 8d's 0.581 TF per generated
        frame x 3 passes x B x (F - 1) = 209 TF at c3 is quoted beside it); frac = algorithmic / time in those launches / peak.
        The other stacks of the step are listed the same way."""
        per = {}
        for k, v in scoped.items():
            scope, lab = k.split('|', 1)
            if not lab.startswith('conv_'):
                continue
            d = per.setdefault(scope, {'ms': 0.0, 'flops': 0.0, 'launches': 0})
            d['ms'] += v['ms']; d['flops'] += v['flops']; d['launches'] += v['launches']
        if 'netG' not in per:
            return None
        g = per['netG']
        tf = g['flops'] / (g['ms'] * 1e-3) / 1e12
        survey = 0.581e12 * 3 * world_batch_frames * (side / 256.0) ** 2 if args.config in ('c1', 'c2', 'c3') else None
        return {"stack": "SPADE decoder conv stack (netG): forward + data gradient + weight gradient launches of one step",
                "algorithmic_tflop_per_step": round(g['flops'] / 1e12, 2),
                "survey_8d_tflop_per_step": (round(survey / 1e12, 2) if survey else None),
                "ms": round(g['ms'], 2), "launches": g['launches'], "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(tf / peak, 4), "peak_basis": peak_basis,
                "other_stacks": {k: {"ms": round(v['ms'], 2), "tflops": round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2), "launches": v['launches'],
                                     "algorithmic_tflop_per_step": round(v['flops'] / 1e12, 2)}
                                 for k, v in sorted(per.items()) if k != 'netG'}}
    world_batch_frames = batch * (frames - 1)          # per rank (the records are rank 0's launches)

    def hbm_roofline(key, kernel, passes):
        if key not in timing:
            return None
        t = timing.pop(key)
        gbs = t['flops'] / (t['ms'] * 1e-3) / 1e9          # the record's work slot holds algorithmic bytes for these kernels
        names = [k for k in PMC if k.startswith(kernel.split('>')[0] + '>')]       # the instantiation named in `kernel`, not a sibling
        pm = PMC.get(names[0]) if names else None
        return {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(gbs / PEAK_HBM_GBS, 4), "launches": t['launches'], "avg_launch_ms": round(t['ms'] / t['launches'], 4),
                "algorithmic_passes": passes,
                "traffic": ((pm['fetch_bytes'] + pm['write_bytes']) if pm else None),
                "traffic_note": (("%s: FETCH %.0f MB + WRITE %.0f MB on shape %s vs %.0f MB algorithmic" % (
                    PMC_PATH, pm['fetch_bytes'] / 1e6, pm['write_bytes'] / 1e6, pm['shape'], pm['algorithmic_bytes'] / 1e6)) if pm else None)}
    # the HBM-bound kernel with the most time: the SPADE pair (norm_0 + norm_s of a block on one input read before its 2x upsampling)
    # when the blocks use it, else the single SPADE modulation
    pair = 'spade_pair_apply' in timing
    roof_hbm_single = hbm_roofline('spade_norm_apply', 'norm_apply_kernel<1, true> (SPADE modulation + LeakyReLU, forward)',
                                   'x, gamma, beta read + y written = 4 tensor passes (the statistics pass is a separate kernel)')
    roof_hbm_single_bwd = hbm_roofline('spade_norm_bwd', 'norm_bwd_reduce_kernel<1, true> + norm_bwd_apply_kernel<1, true> (SPADE modulation backward)',
                                       'reduce: dy, y, x, gamma read; apply: the same four read + dx, dgamma, dbeta written = 11 tensor passes')
    roof_hbm, roof_hbm_bwd = roof_hbm_single, roof_hbm_single_bwd
    if pair:
        roof_hbm = hbm_roofline('spade_pair_apply', 'spade_pair_apply_kernel<2> (two SPADE modulations of one input read before its 2x upsampling)',
                                'x at a quarter of the output size + 2 x (gamma, beta) read, 2 outputs written = 6.25 output-sized passes')
        roof_hbm_bwd = hbm_roofline('spade_pair_bwd', 'spade_pair_bwd_reduce_kernel<2> + spade_pair_bwd_apply_kernel<2>',
                                    'reduce: dy0, dy1, gamma0, gamma1, x (input size) + y_i where slope_i != 1; apply: the same + 2 x (dgamma, dbeta) + dx at '
                                    'input size = 12 + 2 * (#norms with a LeakyReLU) output-sized passes + 0.75 (decoder blocks: 14.75)')
        if roof_hbm is not None:
            roof_hbm['single_spade'] = roof_hbm_single
        if roof_hbm_bwd is not None:
            roof_hbm_bwd['single_spade'] = roof_hbm_single_bwd
    roof = conv_roofline(timing, dt * 1e3 / args.steps) if timing else None     # the records cover the last timed step
    if roof is not None and by_scope:
        roof['spade_stack'] = stack_roofline(by_scope, (PEAK_BF16_MFMA_TFLOPS / 3.0) if args.precision == 'bf16x3' else PEAK_FP32_MFMA_TFLOPS,
                                             ("bf16 dense MFMA 2500 TF / 3 MFMAs per fp32-equivalent product (the stack's <= 4-row and 32-row layers run "
                                              "on exact-fp32 kernels and are priced against the same basis)") if args.precision == 'bf16x3'
                                             else "v_mfma_f32_32x32x2_f32 dense peak")
    roof_fp32 = None
    if fp32_leg is not None:
        roof_fp32 = conv_roofline(fp32_leg['timing'], fp32_leg['ms'])
        if roof_fp32 is not None:
            roof_fp32['spade_stack'] = stack_roofline(fp32_leg['by_scope'], PEAK_FP32_MFMA_TFLOPS, "v_mfma_f32_32x32x2_f32 dense peak")
            roof_fp32["ms_per_step"] = round(fp32_leg['ms'], 2)
            roof_fp32["note"] = "one extra step with --precision fp32 (exact fp32 MFMA everywhere); not part of `value`"
    if roof is not None:
        # flat copies of the nested figures (the driver's record keeps the scalar keys of `roofline` only -- VERDICT r04 #14)
        ss = roof.get('spade_stack') or {}
        roof['spade_stack_frac'] = ss.get('frac')
        roof['spade_stack_tflops'] = ss.get('achieved')
        roof['spade_stack_ms'] = ss.get('ms')
        roof['all_conv_tflops'] = roof['all_conv_kernels']['achieved']
        roof['all_conv_frac'] = round(roof['all_conv_kernels']['achieved'] / roof['peak'], 4)
        roof['hbm_kernel_frac'] = roof_hbm['frac'] if roof_hbm else None
        roof['hbm_kernel_gbs'] = roof_hbm['achieved'] if roof_hbm else None
        roof['exact_fp32_ms_per_step'] = roof_fp32.get('ms_per_step') if roof_fp32 else None
        roof['exact_fp32_frac'] = roof_fp32.get('frac') if roof_fp32 else None
        roof['exact_fp32_spade_stack_frac'] = ((roof_fp32.get('spade_stack') or {}).get('frac')) if roof_fp32 else None
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
    line = {
        "metric": "frames/sec/node (gen+disc fwd+bwd), %s %dx%dx%df" % ("SSv2-shaped" if args.config == 'c5' else "CATER", side, side, frames),
        "value": round(value, 3), "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32 (conv products as 3-term bf16 splits on the bf16 MFMA pipe, fp32 accumulate; everything else fp32)"
                  if args.precision == 'bf16x3' else "f32"), "data": "synthetic CATER-shaped clips (seeded), closed-form random-init weights",
        "config": {"workload": "%s: %s %dx%d, %d frames, batch %d per GPU, full G+D train step, %s%s"
                   % (args.config, "SSv2-shaped" if args.config == 'c5' else "CATER", side, side, frames, batch, ("VGG loss on (random-init VGG19)" if args.vgg_random_init else "VGG loss off"), (", + temporal discriminator (--temporal_D 1)" if args.temporal_D else "") + (", lean tape (--lean_tape 1)" if args.lean_tape else "")), "global_batch": world * batch, "parallelism": "dp%d" % world,
                   "batch_norm": ("statistics synchronised over the %d ranks, the reference's multi-device formula (sync_batchnorm/batchnorm.py:78-93)" % world
                                  if comm.get("sync_bn") else "per-rank statistics, F.batch_norm's formula (the reference's single-device path)")},
        "roofline": roof, "roofline_exact_fp32": roof_fp32, "roofline_hbm": roof_hbm, "roofline_hbm_bwd": roof_hbm_bwd,
        "cpu_baseline": cpu, "comm": comm,
        "extra": {"graph_step_ms": round(graph_ms, 2), "losses_finite": finite,
                  "generated_frames_per_sec": round(world * batch * (frames - 1) * args.steps / dt, 3),
                  "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)},
    }
    assert line['n_gpus'] == args.gpus == comm['rccl_world'], (line['n_gpus'], args.gpus, comm['rccl_world'])
    print(json.dumps(line), flush=True)
    D.shutdown()


if __name__ == '__main__':
    main()
