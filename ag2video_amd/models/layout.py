"""Box -> layout rendering (reference: models/layout.py:28-63 boxes_to_layout and the b x t Python loops that call it,
generator.py:37-54, discriminator.py:321-336)."""
import torch

from .. import ops
from .utils import remove_dummy_objects


def boxes_to_layout(vecs, boxes, H, W=None, keep=None):
    """Single frame: vecs [O, D], boxes [O, 4] (x0, y0, w, h in [0,1]) -> [1, D, H, W]."""
    W = H if W is None else W
    if keep is None:
        keep = torch.ones(vecs.shape[0], dtype=torch.int32, device=vecs.device)
    return ops.layout_render(vecs.unsqueeze(0), boxes.unsqueeze(0), keep.unsqueeze(0), H, W)


def render_clip_layouts(vecs, boxes, objs, vocab, H, W, frame_major=True):
    """All (sample, frame) layouts of a clip in ONE launch.

    vecs [B, F, O, D], boxes [B, F, O, 4], objs [B, O, A] -> list of F tensors [B, D, H, W] (frame_major) so that every
    frame is a contiguous NCHW tensor and the per-frame-step concatenations never touch the other frames.
    Dummy objects (padding / __image__) are masked with ``remove_dummy_objects`` as at generator.py:39-44 -- as a mask
    handed to the kernel instead of boolean indexing (which forces a device->host sync in the reference)."""
    B, Fr, O, D = vecs.shape
    keep = remove_dummy_objects(objs, vocab)                       # [B, O]
    keep = keep.unsqueeze(0).expand(Fr, B, O).reshape(Fr * B, O).to(torch.int32)
    v = vecs.transpose(0, 1).reshape(Fr * B, O, D)
    b = boxes.transpose(0, 1).reshape(Fr * B, O, 4)
    out = ops.layout_render(v, b, keep, H, W)                      # [F*B, D, H, W]
    return list(out.view(Fr, B, D, H, W).unbind(0))


def layout_conv_lowrank(weight, vecs_list, boxes_list, keep, H, W, stride=1, pad=1, residual=None, pool=0, sample_scale=None,
                        img=None, w_img=None, gscale=None):
    """conv2d(concat_f layout_f, weight) WITHOUT materialising the layouts.

    A layout is a sum over objects of (vector) x (separable box weight map) (boxes_to_layout, models/layout.py:28-63), so a
    convolution of it is  sum_objects sum_taps U[obj,k,tap] * weight_map_obj(shifted)  with  U = weight . vector.
    weight [K, n_frames*D, R, R] (the layout channels of the filter, frame-major as in ``seg[:, t-nc:t+1].view(B,-1,H,W)``,
    generator.py:66); vecs_list: n_frames tensors [B, O, D]; boxes_list: n_frames tensors [B, O, 4]; keep [B, O].
    Gradients reach ``weight`` and the vectors through the small GEMM; boxes get none (ground truth / detached).
    sample_scale [B] (optional): the layout term of sample b is multiplied by sample_scale[b] (spectral norm's 1/sigma of that
    sample's frame group when ``weight`` is the raw weight_orig: the term is linear in the filter).
    img [B, IC, H, W] + w_img [K, IC, R, R] (+ gscale [G]): the IMAGE channels of the same convolution, evaluated in the same launch
    (ops.layout_conv_img; stride 1) instead of a dense launch whose output is then read back as ``residual``."""
    K, R = weight.shape[0], weight.shape[2]
    B, O, D = vecs_list[0].shape
    Us = []
    for fi, v in enumerate(vecs_list):
        taps = weight[:, fi * D:(fi + 1) * D].permute(0, 2, 3, 1).reshape(K * R * R, D)      # [(k,r,s), d]
        Us.append(ops.linear(v, taps).view(B, O, K, R * R))
    U = torch.cat(Us, dim=1)
    if sample_scale is not None:
        U = U * sample_scale.view(B, 1, 1, 1)
    boxes = torch.cat(list(boxes_list), dim=1)
    keep_all = keep.to(torch.int32).repeat(1, len(vecs_list))
    if img is not None:
        if residual is None and ops.layout_conv_img_supported(H, W, R, stride, pool, img.shape[1]):
            return ops.layout_conv_img(U, boxes, keep_all, H, W, R, pad, img, w_img, gscale)
        y = ops.conv2d(img, w_img, None, stride, pad, gscale=gscale)
        residual = y if residual is None else residual + y
    return ops.layout_conv(U, boxes, keep_all, H, W, R, stride, pad, residual, pool)
