"""Graph-net recurrences as persistent-kernel programs (ag2video_amd/graph_program.py, csrc/graph_prog.hip) against the per-operator
path they replace (the same modules with AG2V_NO_GRAPH_PROGRAM=1, themselves pinned to the oracle / the reference golden step by
tests/test_ops_gpu.py and tests/test_step_gpu.py): outputs, input gradients and EVERY parameter gradient of Acts2LayoutModel
(reference models/graph_models/model.py:99-174) and of the discriminator's get_obj_vecs (discriminator.py:273-313), at the toy width
and at the default width with CATER-sized graphs (11 object slots, up to 16 edges, 15 recurrence steps).  Tolerance 1e-5 of max: both
are exact-fp32 MFMA with different summation orders."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _opt(hidden):
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.vocab import cater_vocab
    return make_opt(cater_vocab(), ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '2', '--ngf', '8', '--ndf', '8', '--nff', '8',
                                    '--gconv_hidden_dim', str(hidden), '--gpu_ids', '0'])


def _both(fn):
    """fn() with the program path, then with the per-operator path."""
    res = []
    for off in (False, True):
        if off:
            os.environ['AG2V_NO_GRAPH_PROGRAM'] = '1'
        try:
            res.append(fn())
        finally:
            os.environ.pop('AG2V_NO_GRAPH_PROGRAM', None)
    return res


def _close(name, a, b, tol=1e-5):
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err <= tol * max(b.abs().max().item(), 1e-6), (name, err, b.abs().max().item())


@pytest.mark.parametrize("hidden,batch,frames", [(64, 2, 4), (512, 8, 16)])
def test_acts2layout_program_matches_operator_path(hidden, batch, frames, device):
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.models.graph_models.model import Acts2LayoutModel
    from ag2video_amd.weights import closed_form_fill_
    opt = _opt(hidden)
    net = Acts2LayoutModel(opt).to(device)
    closed_form_fill_(net, seed=3, prefix='G.acts_to_objs.module.')
    clip = make_clip_batch(opt.vocab, batch, frames, (64, 64), seed=5)
    objs, boxes, triplets, actions = [t.to(device) for t in (clip[1], clip[2], clip[3], clip[4])]
    wt = torch.randn(batch, frames, objs.shape[1], 128, generator=torch.Generator().manual_seed(1)).to(device)
    wb = torch.randn(batch, frames, objs.shape[1], 4, generator=torch.Generator().manual_seed(2)).to(device)

    def run():
        net.zero_grad(set_to_none=True)
        tv, bp, adata = net(objs, triplets, actions, boxes)
        ((tv * wt).sum() + (bp * wb).sum()).backward()
        torch.cuda.synchronize()
        return tv.detach().clone(), bp.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    (tv, bp, g), (tv_r, bp_r, g_r) = _both(run)
    _close('tvecs', tv, tv_r)
    _close('boxes_pred', bp, bp_r)
    assert set(g) == set(g_r) and len(g) >= 30
    for n in g_r:
        _close(n, g[n], g_r[n], 2e-5)


@pytest.mark.parametrize("hidden,batch,frames", [(64, 2, 4), (512, 8, 16)])
def test_discriminator_graph_net_program_matches_operator_path(hidden, batch, frames, device):
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.models.graph_models.model import Acts2LayoutModel
    from ag2video_amd.models.spade_models.networks.discriminator import MultiscaleActionDiscriminator
    from ag2video_amd.weights import closed_form_fill_
    opt = _opt(hidden)
    D = MultiscaleActionDiscriminator(opt).to(device)
    closed_form_fill_(D, seed=4, prefix='D.')
    clip = make_clip_batch(opt.vocab, batch, frames, (64, 64), seed=6)
    objs, boxes, triplets, actions = [t.to(device) for t in (clip[1], clip[2], clip[3], clip[4])]
    adata = Acts2LayoutModel(opt).to(device).actions_data(triplets, actions)
    r_adata = [x[:, 1:] for x in adata]
    w = torch.randn(batch, frames - 1, objs.shape[1], 128, generator=torch.Generator().manual_seed(3)).to(device)

    def run():
        D.zero_grad(set_to_none=True)
        ov = D.get_obj_vecs(objs, boxes[:, 1:], r_adata)
        (ov * w).sum().backward()
        torch.cuda.synchronize()
        return ov.detach().clone(), {n: p.grad.clone() for n, p in D.named_parameters() if p.grad is not None}
    (ov, g), (ov_r, g_r) = _both(run)
    _close('obj_vecs', ov, ov_r)
    assert set(g) == set(g_r) and len(g) >= 20
    for n in g_r:
        _close(n, g[n], g_r[n], 2e-5)


def test_program_barriers_did_not_time_out(device):
    """The grid barrier is bounded: a run whose workgroups were not co-resident sets a flag instead of hanging."""
    from ag2video_amd import graph_program as gp
    P = gp.Program(device)
    a = torch.randn(40, 64, device=device)
    w = torch.randn(48, 64, device=device)
    out1, out2 = P.new(40, 48), P.new(40, 40)
    P.gemm(P.ext(a), P.ext(w), out1, relu=True)
    P.gemm(out1, out1, out2)                         # depends on every tile of the first operator
    P.finalize().run()
    torch.cuda.synchronize()
    assert not P.timed_out()
    y1 = torch.relu(a @ w.t())
    assert (P.tensor(out1) - y1).abs().max().item() <= 1e-5 * y1.abs().max().item()
    y2 = y1 @ y1.t()
    assert (P.tensor(out2) - y2).abs().max().item() <= 1e-5 * y2.abs().max().item()
