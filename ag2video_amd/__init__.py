"""ag2video_amd -- MI355X-native AG2Video training step (generator + discriminator fwd/bwd).

Host side: PyTorch-ROCm tensors/streams/autograd tape (plumbing) calling a C-ABI
library of hand-written gfx950 HIP kernels (ag2video_amd/csrc, include/ag2v.h).
The module surface mirrors the reference's ``models.meta_models`` / ``LossModel``
(see SURVEY.md 8b).  There is no CPU fallback in this package: ops raise if the
HIP library is missing.
"""
__version__ = "0.1.0"
