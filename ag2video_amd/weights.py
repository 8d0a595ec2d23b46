"""Closed-form, name-keyed parameter filler.

Golden fixtures (tests/golden) are produced by filling the *reference* modules and the
build's modules with the same function, so no weights are ever committed: a tensor's
values depend only on its ``state_dict`` key, its shape and ``seed``.  bench.py uses the
same filler as its "random init of that architecture" (there is no network for
checkpoints).
"""
import zlib

import numpy as np
import torch


def _rs(name, seed):
    return np.random.RandomState((zlib.crc32(name.encode('utf-8')) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def fill_value(name, shape, seed=0):
    """float32 numpy array for state_dict entry ``name`` (int64 zero for num_batches_tracked)."""
    shape = tuple(shape)
    rs = _rs(name, seed)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return np.zeros(shape, np.int64)
    if leaf == 'running_mean':
        return np.zeros(shape, np.float32)
    if leaf == 'running_var':
        return np.ones(shape, np.float32)
    if leaf in ('weight_u', 'weight_v'):
        v = rs.standard_normal(shape).astype(np.float64)
        return (v / max(np.linalg.norm(v), 1e-12)).astype(np.float32)
    if 'att_emb_' in name or 'pred_embeddings' in name or 'acts_embeddings' in name:
        return (0.5 * rs.standard_normal(shape)).astype(np.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        bound = 1.4 * np.sqrt(3.0 / fan_in)
        return rs.uniform(-bound, bound, size=shape).astype(np.float32)
    if leaf == 'bias':
        return rs.uniform(-0.05, 0.05, size=shape).astype(np.float32)
    # 1-D weight: affine scale of a batch norm
    return (1.0 + rs.uniform(-0.1, 0.1, size=shape)).astype(np.float32)


@torch.no_grad()
def closed_form_fill_(module, seed=0, prefix=''):
    """Overwrite every parameter/buffer of ``module`` in place; returns the module."""
    sd = module.state_dict()
    for name, t in sd.items():
        val = torch.from_numpy(fill_value(prefix + name, t.shape, seed))
        t.copy_(val.to(device=t.device, dtype=t.dtype))
    return module
