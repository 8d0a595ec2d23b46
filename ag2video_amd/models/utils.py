"""Helpers of the hot path (reference: models/utils.py:26-51,95-102,113-140)."""
import torch

from .. import ops


def batch_to(batch, device):
    """Host->device copy of a collated batch (models/utils.py:26-51)."""
    out = []
    for obj in batch:
        if torch.is_tensor(obj):
            obj = obj.to(device, non_blocking=True)
        elif isinstance(obj, list) and len(obj) == 0:
            obj = None
        out.append(obj)
    return out


def remove_dummy_objects(objs, vocab):
    """Real-object mask: first attribute is neither padding (0) nor __image__ (models/utils.py:95-102).
    Accepts [O, A] or [B, O, A]; returns a bool mask without any host synchronisation."""
    image_idx = vocab['object_name_to_idx']['__image__']
    first = objs[..., 0]
    return (first != 0) & (first != image_idx)


def resample(image, flow):
    """Bilinear warp by an optical flow, border padding, align_corners=False (models/utils.py:113-124)."""
    return ops.flow_warp(image, flow)
