"""ORACLE (test infrastructure): the reference training iteration (scripts/train.py:440-493) over oracle/model.py.

Used by tests (golden pinning, GPU end-to-end parity), __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import json
import os

import torch

from . import model as M

HERE = os.path.dirname(os.path.abspath(__file__))


class OracleTrainer(object):
    def __init__(self, opt, key_shapes_G, key_shapes_D, wseed, key_shapes_V=None, key_shapes_T=None, dtype=torch.float32):
        """`dtype=torch.float64`: the same fp32 weights and the same fp32 clip, widened, with every operation of the iteration in
        double precision -- the reference point against which the fp32 oracle's OWN rounding is measured (tests/golden/
        make_noise_floor.py -> noise_floor.json) and against which the HIP result is compared beside the fp32 oracle."""
        self.opt = opt
        self.dtype = dtype
        dt = None if dtype == torch.float32 else dtype
        self.G = M.make_state(key_shapes_G, wseed, 'G.', dt)
        self.D = M.make_state(key_shapes_D, wseed, 'D.', dt)
        self.T = M.make_state(key_shapes_T, wseed, 'T.', dt) if key_shapes_T else None      # --temporal_D 1
        self.V = None
        if not opt.no_vgg_loss:           # frozen VGG19 of the perceptual loss (keys 'vgg.slice<k>.<i>.weight|bias')
            self.V = {k: v.detach() for k, v in M.make_state(key_shapes_V, wseed, 'V.', dt).items()}
        lr, b1 = opt.learning_rate, opt.beta1
        graph_params = M.params_of(self.G, 'acts_to_boxes.')
        gen_params = [t for n, t in self.G.items() if t.requires_grad and not n.startswith('acts_to_boxes.')]
        self.opt_graph = torch.optim.Adam(graph_params, lr=lr, betas=(b1, 0.999))
        self.opt_gen = torch.optim.Adam(gen_params, lr=lr, betas=(b1, 0.999))
        self.opt_d = torch.optim.Adam(M.params_of(self.D) + (M.params_of(self.T) if self.T else []), lr=lr, betas=(b1, 0.999))

    def _widen(self, batch):
        return [t.to(self.dtype) if torch.is_tensor(t) and t.is_floating_point() else t for t in batch]

    def _in_dtype(fn):
        """Run a step with torch's default dtype = self.dtype (tensor constructors inside oracle/model.py follow it)."""
        def run(self, batch, *a, **k):
            if self.dtype == torch.float32:
                return fn(self, batch, *a, **k)
            old = torch.get_default_dtype()
            torch.set_default_dtype(self.dtype)
            try:
                return fn(self, self._widen(batch), *a, **k)
            finally:
                torch.set_default_dtype(old)
        return run

    @_in_dtype
    def gd_step(self, batch, apply_updates=True):
        """G step then D step on one clip batch; returns (model_out, G_losses, D_losses)."""
        opt = self.opt
        imgs, objs, boxes, triplets, actions = batch[:5]
        model_out = M.generator_forward(self.G, opt, imgs, objs, triplets, actions, boxes, training=True, use_gt=True)
        G_losses = M.generator_loss(self.D, opt, batch, model_out, V=self.V, PT=self.T)
        G_losses = {k: v.mean() for k, v in G_losses.items()}
        self.opt_gen.zero_grad()
        G_losses['total_loss'].backward()
        self.last_G_grads = {n: (None if t.grad is None else t.grad.clone()) for n, t in self.G.items() if t.requires_grad}
        if apply_updates:
            self.opt_gen.step()
        D_losses = M.discriminator_loss(self.D, opt, batch, model_out, PT=self.T)
        D_losses = {k: v.mean() for k, v in D_losses.items()}
        self.opt_d.zero_grad()
        D_losses['total_img_loss'].backward()
        self.last_D_grads = {n: (None if t.grad is None else t.grad.clone()) for n, t in self.D.items() if t.requires_grad}
        if self.T:
            self.last_T_grads = {n: (None if t.grad is None else t.grad.clone()) for n, t in self.T.items() if t.requires_grad}
        if apply_updates:
            self.opt_d.step()
        return model_out, G_losses, D_losses

    @_in_dtype
    def graph_step(self, batch, apply_updates=True):
        imgs, objs, boxes, triplets, actions = batch[:5]
        boxes_pred = M.generator_forward(self.G, self.opt, imgs, objs, triplets, actions, boxes, training=True, graph_only=True)
        L = M.graph_loss(self.opt, batch, boxes_pred)
        L = {k: v.mean() for k, v in L.items()}
        self.opt_graph.zero_grad()
        L['total_loss'].backward()
        self.last_graph_grads = {n: (None if t.grad is None else t.grad.clone()) for n, t in self.G.items()
                                 if t.requires_grad and n.startswith('acts_to_boxes.')}
        if apply_updates:
            self.opt_graph.step()
        return boxes_pred, L


def toy_key_shapes():
    """state_dict key -> shape tables of the toy configuration (written by tests/golden/make_golden.py)."""
    p = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'toy64_state_keys.json')
    ks = json.load(open(p))
    return ks['model'], ks['d_img'], ks['gans_model']
