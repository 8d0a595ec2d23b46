"""Parameter-holding building blocks whose forward runs on the HIP operators (ag2video_amd.ops).

They subclass the torch.nn parameter containers only to inherit parameter registration, default initialisation and
``state_dict`` naming (the reference relies on PyTorch's default init everywhere except GraphTripleConv's Linear layers,
models/graph_models/graph.py:10-13); no torch.nn.functional arithmetic is used.
Mirrors models/layers.py:6-25 (build_mlp) of the reference.
"""
import torch
import torch.nn as nn

from .. import ops


class Linear(nn.Linear):
    def forward(self, x, relu=False):
        return ops.linear(x, self.weight, self.bias, relu=relu)


class Embedding(nn.Embedding):
    def forward(self, idx):
        return ops.embedding(idx, self.weight)


class Conv2d(nn.Conv2d):
    """Plain convolution (square kernel/stride/padding)."""

    def forward(self, x, residual=None):
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], residual)


class FusedMLP(nn.Sequential):
    """nn.Sequential of Linear / ReLU modules (same child indices as build_mlp => same state_dict keys) whose forward
    fuses every Linear with the ReLU that follows it into one GEMM epilogue."""

    def forward(self, x):
        mods = list(self.children())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, Linear):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = m(x, relu=relu)
                i += 2 if relu else 1
            elif isinstance(m, nn.ReLU):
                x = ops.leaky_relu(x, 0.0)
                i += 1
            else:
                raise RuntimeError("FusedMLP: unsupported child %r" % type(m))
        return x


def build_mlp(dim_list, activation='relu', batch_norm='none', dropout=0, final_nonlinearity='relu'):
    """models/layers.py:6-25 for the only configuration on the hot path (relu, no batch norm, no dropout)."""
    if activation != 'relu' or batch_norm not in ('none', None) or dropout:
        raise NotImplementedError("build_mlp: only relu / no normalisation / no dropout is on the hot path")
    layers = []
    for i in range(len(dim_list) - 1):
        layers.append(Linear(dim_list[i], dim_list[i + 1]))
        if i != len(dim_list) - 2:
            layers.append(nn.ReLU())
    if final_nonlinearity is not None:
        layers.append(nn.ReLU())
    return FusedMLP(*layers)
