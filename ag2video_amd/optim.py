"""Fused Adam over flat parameter / gradient buffers.

The reference uses three torch.optim.Adam instances (scripts/train.py:365-368, meta_models.py:70-72).  Here every
optimizer owns ONE flat fp32 buffer for its parameters, one for their gradients and two for the Adam moments; each
nn.Parameter (and its .grad) is a view into those buffers.  The update is a single HIP kernel launch (csrc/pointwise.hip
adam_kernel), zero_grad is one memset, and the data-parallel gradient all-reduce runs directly on the flat gradient
buffer (no bucket copies).  state_dict()/load_state_dict() keep torch.optim.Adam's layout (per-parameter ``step``,
``exp_avg``, ``exp_avg_sq``), so reference checkpoints ('optim_state_gen', 'd_img_optim_state', ...) remain loadable.
"""
import torch

from . import _lib
from .ops import _ptr, _stream, bump_param_epoch


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = [p for p in params]
        if not params:
            raise ValueError("FusedAdam: empty parameter list")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        ps = self.param_groups[0]['params']
        dev = ps[0].device
        # (the flat layout itself is device-agnostic -- tests/test_distributed_cpu.py drives engine.gd_step's two-phase all-reduce over
        # it under gloo with a test-only subclass that overrides step(); the update below exists as a HIP kernel only)
        self._sizes = [p.numel() for p in ps]
        # every parameter starts on a 16-byte boundary (kernels read weights with 16-byte loads: the graph nets' GEMM, the weight
        # packing); the <= 3 padding elements between parameters stay 0 in all four buffers (Adam maps 0, 0, 0, 0 to 0)
        offs, total = [], 0
        for n in self._sizes:
            offs.append(total)
            total += (n + 3) & ~3
        self.flat_param = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(total, device=dev, dtype=torch.float32)
        self._step = 0
        self._views = []
        with torch.no_grad():
            for p, n, off in zip(ps, self._sizes, offs):
                if p.dtype != torch.float32:
                    raise RuntimeError("FusedAdam: fp32 parameters only")
                pv = self.flat_param[off:off + n].view_as(p)
                pv.copy_(p.detach())
                p.data = pv
                gv = self.flat_grad[off:off + n].view_as(p)
                p.grad = gv
                self._views.append((off, n, gv))

    def _adopt_grads(self):
        """Make every .grad the flat view again (someone may have set it to None or replaced it)."""
        for p, (off, n, gv) in zip(self.param_groups[0]['params'], self._views):
            if p.grad is gv:
                continue
            if p.grad is not None:
                gv.copy_(p.grad)        # a replaced gradient tensor: fold it into the flat buffer
            p.grad = gv

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        for p, (off, n, gv) in zip(self.param_groups[0]['params'], self._views):
            p.grad = gv

    @torch.no_grad()
    def step(self, closure=None):
        if self.flat_param.device.type != 'cuda':
            raise RuntimeError("FusedAdam.step runs on the HIP device only (no CPU fallback); got %s" % self.flat_param.device)
        loss = closure() if closure is not None else None
        self._adopt_grads()
        g = self.param_groups[0]
        self._step += 1
        _lib.call('ag2v_adam_step', _ptr(self.flat_param), _ptr(self.flat_grad), _ptr(self.flat_m), _ptr(self.flat_v),
                  self.flat_param.numel(), float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                  int(self._step), _stream())
        bump_param_epoch()     # the kernel wrote the parameters through raw pointers: invalidate derived-tensor caches
        return loss

    # ---- torch.optim.Adam-compatible (de)serialisation
    def state_dict(self):
        state = {}
        for i, (off, n, gv) in enumerate(self._views):
            state[i] = {'step': torch.tensor(float(self._step)),
                        'exp_avg': self.flat_m[off:off + n].view(gv.shape).clone(),
                        'exp_avg_sq': self.flat_v[off:off + n].view(gv.shape).clone()}
        g = self.param_groups[0]
        groups = [{'lr': g['lr'], 'betas': g['betas'], 'eps': g['eps'], 'weight_decay': 0, 'amsgrad': False,
                   'params': list(range(len(self._views)))}]
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        steps = []
        for i, (off, n, gv) in enumerate(self._views):
            st = sd['state'].get(i, sd['state'].get(str(i)))
            if st is None:
                continue
            self.flat_m[off:off + n].copy_(st['exp_avg'].reshape(-1).to(self.flat_m.device))
            self.flat_v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1).to(self.flat_v.device))
            steps.append(int(float(st['step'])))
        if steps:
            self._step = max(steps)
        pg = sd.get('param_groups', [{}])[0]
        for k in ('lr', 'betas', 'eps'):
            if k in pg:
                self.param_groups[0][k] = pg[k]
