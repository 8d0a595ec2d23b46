"""Conditioning of the graph net's gradients: the per-operator path with its two GEMM kernels (different summation orders)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    import test_graph_program_gpu as T
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.models.graph_models.model import Acts2LayoutModel
    from ag2video_amd.weights import closed_form_fill_
    dev = torch.device('cuda:0')
    opt = T._opt(512)
    net = Acts2LayoutModel(opt).to(dev)
    closed_form_fill_(net, seed=3, prefix='G.acts_to_objs.module.')
    clip = make_clip_batch(opt.vocab, 8, 16, (64, 64), seed=5)
    objs, boxes, triplets, actions = [t.to(dev) for t in (clip[1], clip[2], clip[3], clip[4])]
    wt = torch.randn(8, 16, objs.shape[1], 128, generator=torch.Generator().manual_seed(1)).to(dev)
    wb = torch.randn(8, 16, objs.shape[1], 4, generator=torch.Generator().manual_seed(2)).to(dev)
    tv, bp, _ = net(objs, triplets, actions, boxes)
    ((tv * wt).sum() + (bp * wb).sum()).backward()
    torch.save({n: p.grad.cpu() for n, p in net.named_parameters() if p.grad is not None}, sys.argv[1])
else:
    import torch
    outs = []
    for i, env in enumerate(({'AG2V_NO_GRAPH_PROGRAM': '1'}, {'AG2V_NO_GRAPH_PROGRAM': '1', 'AG2V_GEMM_ONESHOT': '0'}, {})):
        f = '/tmp/cond%d.pt' % i
        subprocess.run([sys.executable, os.path.abspath(__file__), f], env=dict(os.environ, **env), check=True)
        outs.append(torch.load(f))
    a, b, c = outs
    worst = lambda x, y: max(((x[n] - y[n]).abs().max() / y[n].abs().max()).item() for n in y)
    l2 = lambda x, y: max(((x[n] - y[n]).norm() / y[n].norm()).item() for n in y)
    print('operators(oneshot GEMM) vs operators(K-loop GEMM): worst max-rel %.2e  worst rel-L2 %.2e' % (worst(b, a), l2(b, a)))
    print('program vs operators(oneshot GEMM):               worst max-rel %.2e  worst rel-L2 %.2e' % (worst(c, a), l2(c, a)))
    print('program vs operators(K-loop GEMM):                worst max-rel %.2e  worst rel-L2 %.2e' % (worst(c, b), l2(c, b)))
