"""Child of tests/test_distributed_gpu.py::test_sync_bn_two_ranks_distinct_data: one of TWO ranks (torchrun) that share the box's one
GPU.  RCCL refuses two ranks on one device, so the process group is gloo and the all-reduces of GPU tensors are staged through the
host HERE (a test-side wrapper of torch.distributed.all_reduce; the product calls the same function with device tensors under RCCL).

What it pins: with RANK-DISTINCT data, every fused synchronised-batch-norm path of ag2video_amd.ops -- batchnorm_act
(mode 0), batchnorm_act_pyramid, spade_norm_fused, spade_pair (up = 1, 2) -- against an fp64 evaluation of the reference's multi-device
formula (sync_batchnorm/batchnorm.py:80-93,105-145: statistics over ALL ranks' samples of a frame group, inv_std = clamp(biased var,
eps)^-1/2, running_var fed with the unbiased variance) on the concatenated batch: outputs, d(x), d(gamma | beta), the per-rank affine
gradients and the running buffers.  A wrong column set in ops._sync_bwd_sums or a wrong global count passes a one-rank run (the
all-reduce is the identity there) and fails here.
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main(out_path):
    import torch.distributed as dist
    from ag2video_amd import distributed as D
    from ag2video_amd import ops
    rank, world, _ = D.init_from_env(backend='gloo')
    assert world == 2 and dist.get_backend() == 'gloo' and D.is_dist()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    inner = dist.all_reduce
    n_calls = [0]

    def staged(t, *a, **k):                      # device tensor -> host -> gloo -> device
        n_calls[0] += 1
        if t.is_cuda:
            h = t.detach().cpu()
            inner(h, *a, **k)
            t.copy_(h)
            return None
        return inner(t, *a, **k)
    dist.all_reduce = staged
    D.set_sync_bn(True)
    assert D.sync_bn()
    N, C, H, W, groups, eps, mom = 4, 8, 16, 16, 2, 1e-5, 0.1
    Ng = N // groups

    def data(seed, *shape, scale=1.0):
        g = torch.Generator().manual_seed(seed)
        return torch.randn(*shape, generator=g) * scale

    def make_x(r, h, w, tag):
        x = data(1000 * tag + r, N, C, h, w, scale=1.5) + torch.linspace(-1, 1, C).view(1, C, 1, 1) * (1.0 + r)
        x[:, C - 1] *= 1e-4                      # a channel whose variance sits below eps: the clamp branch of the formula
        return x

    def ref_norm(x_all, rm, rv):
        """x_all [world, N, C, h, w] fp64 -> xhat (global statistics per frame group), running buffers after the groups in order."""
        outs = []
        for g in range(groups):
            xg = x_all[:, g * Ng:(g + 1) * Ng]
            n = xg.numel() // C
            mean = xg.mean(dim=(0, 1, 3, 4), keepdim=True)
            var = ((xg - mean) ** 2).mean(dim=(0, 1, 3, 4), keepdim=True)
            with torch.no_grad():
                rm = (1 - mom) * rm + mom * mean.reshape(C)
                rv = (1 - mom) * rv + mom * (var.reshape(C) * n / (n - 1))
            outs.append((xg - mean) * var.clamp(min=eps).pow(-0.5))
        return torch.cat(outs, dim=1), rm, rv

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))

    res = {}
    rm0, rv0 = torch.linspace(-0.5, 0.5, C).double(), torch.linspace(0.5, 2.0, C).double()

    # ---------------------------------------------------------------- batchnorm_act (mode 0) and batchnorm_act_pyramid
    sizes = ((16, 16), (8, 8), (4, 4))
    for name in ('batchnorm_act', 'batchnorm_act_pyramid'):
        xs_all = [make_x(r, H, W, 1) for r in range(world)]
        w0, b0 = torch.linspace(0.5, 1.5, C), torch.linspace(-1, 1, C)
        n_out = len(sizes) if name.endswith('pyramid') else 1
        cots = [[data(77 + 10 * r + k, N, C, *(sizes[k] if n_out > 1 else (H, W))) for k in range(n_out)] for r in range(world)]
        # GPU, this rank
        x = xs_all[rank].to(dev).requires_grad_(True)
        w, b = w0.to(dev).requires_grad_(True), b0.to(dev).requires_grad_(True)
        rm, rv = rm0.float().to(dev), rv0.float().to(dev)
        if n_out > 1:
            assert ops.batchnorm_act_pyramid_supported(x.shape, sizes)
            outs = ops.batchnorm_act_pyramid(x, w, b, rm, rv, sizes, True, mom, eps, 0.2, groups=groups)
        else:
            outs = (ops.batchnorm_act(x, w, b, rm, rv, True, mom, eps, 0.2, groups=groups),)
        sum((o * c.to(dev)).sum() for o, c in zip(outs, cots[rank])).backward()
        # fp64, all ranks
        xa = torch.stack(xs_all).double().requires_grad_(True)
        wr = [w0.double().clone().requires_grad_(True) for _ in range(world)]
        br = [b0.double().clone().requires_grad_(True) for _ in range(world)]
        xhat, rm_r, rv_r = ref_norm(xa, rm0, rv0)
        loss, outs_r = 0.0, []
        for r in range(world):
            y = F.leaky_relu(xhat[r] * wr[r].view(1, C, 1, 1) + br[r].view(1, C, 1, 1), 0.2)
            lv = [F.interpolate(y, size=s, mode='nearest') for s in sizes] if n_out > 1 else [y]
            outs_r.append(lv)
            loss = loss + sum((o * c.double()).sum() for o, c in zip(lv, cots[r]))
        loss.backward()
        res[name] = {'y': max(rel(o, orf) for o, orf in zip(outs, outs_r[rank])), 'dx': rel(x.grad, xa.grad[rank]),
                     'dw': rel(w.grad, wr[rank].grad), 'db': rel(b.grad, br[rank].grad),
                     'rm': rel(rm, rm_r), 'rv': rel(rv, rv_r)}

    # ---------------------------------------------------------------- spade_norm_fused
    xs_all = [make_x(r, H, W, 2) for r in range(world)]
    gbs = [data(300 + r, N, 2 * C, H, W, scale=0.5) for r in range(world)]
    cots1 = [data(400 + r, N, C, H, W) for r in range(world)]
    x = xs_all[rank].to(dev).requires_grad_(True)
    gb = gbs[rank].to(dev).requires_grad_(True)
    rm, rv = rm0.float().to(dev), rv0.float().to(dev)
    y = ops.spade_norm_fused(x, gb, rm, rv, True, mom, eps, 0.2, groups=groups)
    (y * cots1[rank].to(dev)).sum().backward()
    xa = torch.stack(xs_all).double().requires_grad_(True)
    ga = torch.stack(gbs).double().requires_grad_(True)
    xhat, rm_r, rv_r = ref_norm(xa, rm0, rv0)
    ys = F.leaky_relu(xhat * (1 + ga[:, :, :C]) + ga[:, :, C:], 0.2)
    sum((ys[r] * cots1[r].double()).sum() for r in range(world)).backward()
    res['spade_norm_fused'] = {'y': rel(y, ys[rank]), 'dx': rel(x.grad, xa.grad[rank]), 'dgb': rel(gb.grad, ga.grad[rank]),
                               'rm': rel(rm, rm_r), 'rv': rel(rv, rv_r)}

    # ---------------------------------------------------------------- spade_pair, up = 1 and 2
    for up in (1, 2):
        hs, ws_ = H // up, W // up
        assert ops.spade_pair_supported(H, W, up)
        xs_all = [make_x(r, hs, ws_, 3 + up) for r in range(world)]
        g0 = [data(500 + 10 * up + r, N, 2 * C, H, W, scale=0.5) for r in range(world)]
        g1 = [data(600 + 10 * up + r, N, 2 * C, H, W, scale=0.5) for r in range(world)]
        c0 = [data(700 + 10 * up + r, N, C, H, W) for r in range(world)]
        c1 = [data(800 + 10 * up + r, N, C, H, W) for r in range(world)]
        x = xs_all[rank].to(dev).requires_grad_(True)
        a0, a1 = g0[rank].to(dev).requires_grad_(True), g1[rank].to(dev).requires_grad_(True)
        rma, rva = rm0.float().to(dev), rv0.float().to(dev)
        rmb, rvb = (rm0 * 0.5).float().to(dev), (rv0 * 2.0).float().to(dev)
        y0, y1 = ops.spade_pair(x, a0, a1, rma, rva, rmb, rvb, up=up, momentum=mom, eps=eps, slope0=0.2, slope1=1.0, groups=groups)
        ((y0 * c0[rank].to(dev)).sum() + (y1 * c1[rank].to(dev)).sum()).backward()
        xa = torch.stack(xs_all).double().requires_grad_(True)
        A0, A1 = torch.stack(g0).double().requires_grad_(True), torch.stack(g1).double().requires_grad_(True)
        xu = xa.reshape(world * N, C, hs, ws_)
        xu = F.interpolate(xu, scale_factor=2, mode='nearest') if up == 2 else xu
        xu = xu.reshape(world, N, C, H, W)
        xhat, rma_r, rva_r = ref_norm(xu, rm0, rv0)
        _, rmb_r, rvb_r = ref_norm(xu.detach(), rm0 * 0.5, rv0 * 2.0)
        r0 = F.leaky_relu(xhat * (1 + A0[:, :, :C]) + A0[:, :, C:], 0.2)
        r1 = xhat * (1 + A1[:, :, :C]) + A1[:, :, C:]
        sum((r0[r] * c0[r].double()).sum() + (r1[r] * c1[r].double()).sum() for r in range(world)).backward()
        res['spade_pair_up%d' % up] = {'y': max(rel(y0, r0[rank]), rel(y1, r1[rank])), 'dx': rel(x.grad, xa.grad[rank]),
                                       'dgb': max(rel(a0.grad, A0.grad[rank]), rel(a1.grad, A1.grad[rank])),
                                       'rm': max(rel(rma, rma_r), rel(rmb, rmb_r)), 'rv': max(rel(rva, rva_r), rel(rvb, rvb_r))}
    res['all_reduce_calls'] = n_calls[0]
    torch.cuda.synchronize()
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        json.dump({'world': world, 'ranks': gathered}, open(out_path, 'w'))
    dist.all_reduce = inner
    D.set_sync_bn(False)
    D.shutdown()


if __name__ == '__main__':
    main(sys.argv[1])
