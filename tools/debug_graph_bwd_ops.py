"""Verifies every operator of a graph-net BACKWARD program against torch arithmetic on the program's own tensors (default width, c3 graphs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import test_graph_program_gpu as T
from ag2video_amd import graph_program as gp
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
torch.cuda.synchronize()
pl = [p for pool in gp._PLANS.values() for p in pool if p.bwd is not None][0]
F, P, f, g = pl.fwd, pl.bwd, pl.f, pl.g
FT = lambda n: F.tensor(f[n]).double()
GT = lambda n: P.tensor(g[n]).double()
E, Hd, Dp, Dout, Dins, dp0 = pl.dims
S, BO, BT, O, Tn = pl.S, pl.BO, pl.BT, pl.O, pl.T
params = [p.double() for p in pl._params]
W1, W2 = params[0], params[1]
gcs = [params[2 + 8 * l: 10 + 8 * l] for l in range(3)]
box = params[26:30]
i0, i1, valid = pl.i0.long(), pl.i1.long(), pl.valid.bool()


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item()


d_ov, d_boxes = pl.d_ov.double(), pl.d_boxes.double()
gbox, din0 = GT('gbox'), GT('din0')
for s in range(S - 1, -1, -1):
    ref = d_boxes[s] + (gbox[s + 1] + din0[s + 1][:, E:E + 4] if s < S - 1 else 0)
    e = rel(gbox[s], ref)
    if e > 1e-5: print('step', s, 'gbox', e)
    dbh = gbox[s] @ box[2]
    e = rel(GT('dbh')[s], dbh)
    if e > 1e-5: print('step', s, 'dbh', e)
    top = (GT('dbh')[s] * (FT('bh')[s] > 0)) @ box[0] + d_ov[s]
    e = rel(GT('dov3')[s], top)
    if e > 1e-5: print('step', s, 'dov3', e)
    for l in (2, 1, 0):
        n1a_w, _, n1b_w, _, n2a_w, _, n2b_w, _ = gcs[l]
        e = rel(GT('da2%d' % l)[s], (GT('dov%d' % (l + 1))[s] * (FT('ov%d' % (l + 1))[s] > 0)) @ n2b_w)
        if e > 1e-5: print('step', s, 'da2', l, e)
        e = rel(GT('dpool%d' % l)[s], (GT('da2%d' % l)[s] * (FT('a2%d' % l)[s] > 0)) @ n2a_w)
        if e > 1e-5: print('step', s, 'dpool', l, e)
        cnt = FT('cnt%d' % l)[s].view(-1)
        dpool = GT('dpool%d' % l)[s]
        v = valid[s].double().view(-1, 1)
        inv = torch.where(cnt > 0, 1.0 / cnt.clamp(min=1), torch.zeros_like(cnt))
        dnt = torch.zeros(BT, 2 * Hd + Dp, dtype=torch.float64, device=dev)
        dnt[:, :Hd] = dpool[i0[s]] * inv[i0[s]].view(-1, 1) * v
        dnt[:, Hd + Dp:] = dpool[i1[s]] * inv[i1[s]].view(-1, 1) * v
        if l < 2:
            dnt[:, Hd:Hd + Dp] = GT('dcur%d' % (l + 1))[s][:, Dins[l + 1]:Dins[l + 1] + Dp]
        e = rel(GT('dnt%d' % l)[s], dnt)
        if e > 1e-5: print('step', s, 'dnt(pool_bwd)', l, e)
        e = rel(GT('da1%d' % l)[s], (GT('dnt%d' % l)[s] * (FT('nt%d' % l)[s] > 0)) @ n1b_w)
        if e > 1e-5: print('step', s, 'da1', l, e)
        e = rel(GT('dcur%d' % l)[s], (GT('da1%d' % l)[s] * (FT('a1%d' % l)[s] > 0)) @ n1a_w)
        if e > 1e-5: print('step', s, 'dcur', l, e)
        dcur = GT('dcur%d' % l)[s]
        dpi = dp0 if l == 0 else Dp
        dov = torch.zeros(BO, Dins[l], dtype=torch.float64, device=dev)
        dov.index_add_(0, i0[s], dcur[:, :Dins[l]])
        dov.index_add_(0, i1[s], dcur[:, Dins[l] + dpi:])
        e = rel(GT('dov%d' % l)[s], dov)
        if e > 1e-5: print('step', s, 'dov(gather_bwd)', l, e)
    e = rel(GT('dh1')[s], (GT('dov0')[s] * (FT('ov0')[s] > 0)) @ W2)
    if e > 1e-5: print('step', s, 'dh1', e)
    e = rel(din0[s], (GT('dh1')[s] * (FT('h1')[s] > 0)) @ W1)
    if e > 1e-5: print('step', s, 'din0', e)
print('checked', S, 'steps')
# weight-gradient stage: torch fp64 from the program's stacks against what the backward handed to autograd
pg = dict(net.named_parameters())
def wref(A, dY, Y):
    dZ = dY * (Y > 0) if Y is not None else dY
    return torch.einsum('smn,smk->nk', dZ, A), dZ.sum(dim=(0, 1))
checks = [('box_net.2', FT('bh'), GT('gbox'), None), ('box_net.0', FT('ov3'), GT('dbh'), FT('bh')),
          ('gconvs.2.net2.2', FT('a22'), GT('dov3'), FT('ov3')), ('gconvs.0.net1.0', FT('cur0'), GT('da10'), FT('a10')),
          ('obj_vecs_net.2', FT('h1'), GT('dov0'), FT('ov0')), ('obj_vecs_net.0', FT('in0'), GT('dh1'), FT('h1'))]
for name, A, dY, Y in checks:
    w_ref, b_ref = wref(A, dY, Y)
    print('%-18s weight grad vs fp64 recomputation %.2e' % (name, rel(pg[name + '.weight'].grad.double(), w_ref)),
          ('bias %.2e' % rel(pg[name + '.bias'].grad.double(), b_ref)) if (name + '.bias') in pg else '')
# inputs of the backward program: gradients of the outputs as autograd delivered them
B_, Fr = 8, 16
exp_dov = wt[:, 1:].permute(1, 0, 2, 3).reshape(S, BO, 128).double()
exp_db = wb[:, 1:].permute(1, 0, 2, 3).reshape(S, BO, 4).double()
print('d_ov slot vs expected %.2e   d_boxes slot vs expected %.2e' % (rel(d_ov, exp_dov), rel(d_boxes, exp_db)))
# the program's inputs against the model's tensors
emb = net.attribute_embedding(objs).reshape(BO, -1).double()
print('x0 slot vs attribute embedding %.2e' % rel(pl.x0.double(), emb))
print('boxes_in slot vs boxes[:, 0] %.2e' % rel(pl.boxes_in.double(), boxes[:, 0].reshape(BO, 4).double()))
for l in range(3):
    for s in (0, 7, 14):
        cr = torch.zeros(BO, dtype=torch.float64, device=dev)
        v = valid[s].double()
        cr.index_add_(0, i0[s], v); cr.index_add_(0, i1[s], v)
        c = FT('cnt%d' % l)[s].view(-1)
        print('counts layer %d step %d: max abs diff %.1f (max count %.0f)' % (l, s, (c - cr).abs().max().item(), cr.max().item()))
# pooled forward against the formula
for l in range(3):
    s = 3
    nt = FT('nt%d' % l)[s]
    v = valid[s].double().view(-1, 1)
    pr = torch.zeros(BO, Hd, dtype=torch.float64, device=dev)
    pr.index_add_(0, i0[s], nt[:, :Hd] * v); pr.index_add_(0, i1[s], nt[:, Hd + Dp:] * v)
    cr = torch.zeros(BO, dtype=torch.float64, device=dev); cr.index_add_(0, i0[s], v.view(-1)); cr.index_add_(0, i1[s], v.view(-1))
    pr = torch.where(cr.view(-1, 1) > 0, pr / cr.clamp(min=1).view(-1, 1), pr)
    print('pooled layer %d: %.2e' % (l, rel(FT('pool%d' % l)[s], pr)))
# ---- independent fp64 autograd of the recurrence from the plan's inputs; compare every intermediate gradient with the program's stacks
x0 = pl.x0.double().requires_grad_(True)
pvs = pl.pv.double().requires_grad_(True)
P64 = [p.clone().requires_grad_(True) for p in params]
W1_, W2_ = P64[0], P64[1]
gcs_ = [P64[2 + 8 * l: 10 + 8 * l] for l in range(3)]
box_ = P64[26:30]
bprev = pl.boxes_in.double()
inter = {}
outs_ov, outs_bx = [], []
relu = torch.relu
for s in range(S):
    in0 = torch.cat([x0, bprev], dim=1); in0.retain_grad(); inter[('in0', s)] = in0
    h1 = relu(in0 @ W1_.t()); ov = relu(h1 @ W2_.t()); ov.retain_grad(); inter[('ov0', s)] = ov
    pv = pvs[s]
    for l in range(3):
        n1a_w, n1a_b, n1b_w, n1b_b, n2a_w, n2a_b, n2b_w, n2b_b = gcs_[l]
        cur = torch.cat([ov[i0[s]], pv, ov[i1[s]]], dim=1); cur.retain_grad(); inter[('cur%d' % l, s)] = cur
        a1 = relu(cur @ n1a_w.t() + n1a_b); nt = relu(a1 @ n1b_w.t() + n1b_b); nt.retain_grad(); inter[('nt%d' % l, s)] = nt
        v = valid[s].double().view(-1, 1)
        pr = torch.zeros(BO, Hd, dtype=torch.float64, device=dev).index_add(0, i0[s], nt[:, :Hd] * v).index_add(0, i1[s], nt[:, Hd + Dp:] * v)
        cr = torch.zeros(BO, dtype=torch.float64, device=dev).index_add(0, i0[s], v.view(-1)).index_add(0, i1[s], v.view(-1))
        pooled = torch.where(cr.view(-1, 1) > 0, pr / cr.clamp(min=1).view(-1, 1), pr); pooled.retain_grad(); inter[('pool%d' % l, s)] = pooled
        a2 = relu(pooled @ n2a_w.t() + n2a_b); ov = relu(a2 @ n2b_w.t() + n2b_b); ov.retain_grad(); inter[('ov%d' % (l + 1), s)] = ov
        pv = nt[:, Hd:Hd + Dp]
    bh = relu(ov @ box_[0].t() + box_[1]); bnew = bh @ box_[2].t() + box_[3] + bprev; bnew.retain_grad(); inter[('boxes', s)] = bnew
    outs_ov.append(ov); outs_bx.append(bnew); bprev = bnew
loss = (torch.stack(outs_ov) * d_ov).sum() + (torch.stack(outs_bx) * d_boxes).sum()
loss.backward()
print('fp64 forward vs program forward: ov3 %.2e boxes %.2e' % (rel(FT('ov3'), torch.stack(outs_ov).detach()), rel(FT('boxes'), torch.stack(outs_bx).detach())))
names = {'in0': 'din0', 'ov0': 'dov0', 'boxes': 'gbox'}
for l in range(3):
    names.update({'cur%d' % l: 'dcur%d' % l, 'nt%d' % l: None, 'pool%d' % l: 'dpool%d' % l, 'ov%d' % (l + 1): 'dov%d' % (l + 1)})
for s in (S - 1, S - 2, 7, 0):
    row = []
    for k in ('boxes', 'ov3', 'pool2', 'cur2', 'ov2', 'pool1', 'cur1', 'ov1', 'pool0', 'cur0', 'ov0', 'in0'):
        gname = names[k]
        if gname is None:
            continue
        row.append('%s %.1e' % (gname, rel(GT(gname)[s], inter[(k, s)].grad)))
    print('step %2d:' % s, '  '.join(row))
# who is the outlier?  independent fp64 recurrence vs the program path's parameter gradients vs the per-operator path's
pnames = ['obj_vecs_net.0.weight', 'obj_vecs_net.2.weight']
for l in range(3):
    pnames += ['gconvs.%d.net1.0.weight' % l, 'gconvs.%d.net1.0.bias' % l, 'gconvs.%d.net1.2.weight' % l, 'gconvs.%d.net1.2.bias' % l,
               'gconvs.%d.net2.0.weight' % l, 'gconvs.%d.net2.0.bias' % l, 'gconvs.%d.net2.2.weight' % l, 'gconvs.%d.net2.2.bias' % l]
pnames += ['box_net.0.weight', 'box_net.0.bias', 'box_net.2.weight', 'box_net.2.bias']
prog = {n: pg[n].grad.double().clone() for n in pnames}
os.environ['AG2V_NO_GRAPH_PROGRAM'] = '1'
net.zero_grad(set_to_none=True)
tv, bp, _ = net(objs, triplets, actions, boxes)
((tv * wt).sum() + (bp * wb).sum()).backward()
opsg = {n: pg[n].grad.double().clone() for n in pnames}
ind = {n: P64[i].grad for i, n in enumerate(pnames)}
print('no grad in the fp64 run:', [n for n in pnames if ind[n] is None])
pnames = [n for n in pnames if ind[n] is not None]
worst = lambda A, Bv: max(rel(A[n], Bv[n]) for n in pnames)
print('program vs independent fp64: %.2e   operators vs independent fp64: %.2e   program vs operators: %.2e' % (worst(prog, ind), worst(opsg, ind), worst(prog, opsg)))
