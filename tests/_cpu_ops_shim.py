"""TEST-ONLY: lets the CPU suite exercise the product's *host* orchestration (ag2video_amd/models, engine) without a GPU by
temporarily substituting the oracle's CPU restatements for the HIP operators.  The product package itself has no such
switch; this context manager lives in tests/ and is never used on the GPU path."""
import contextlib

OPS = ['conv2d', 'conv2d_shared_input', 'spectral_weight', 'spectral_scales', 'SpectralBank', 'batchnorm_act', 'spade_norm', 'spade_norm_fused', 'spade_pair', 'spade_pair_supported', 'instnorm_lrelu', 'layout_render', 'layout_conv', 'layout_conv_img', 'layout_conv_img_supported', 'flow_warp', 'conv_split_input_ok', 'batchnorm_act_pyramid', 'batchnorm_act_pyramid_supported',
       'avgpool3s2', 'maxpool2', 'nearest_resize', 'nearest_pyramid', 'upsample2x', 'linear', 'triple_gather', 'triple_pool', 'embedding', 'leaky_relu',
       'tanh', 'l1_mean', 'hinge_mean']


@contextlib.contextmanager
def oracle_ops():
    from ag2video_amd import ops
    from oracle import kernels as K
    saved = {n: getattr(ops, n) for n in OPS}
    try:
        for n in OPS:
            setattr(ops, n, getattr(K, n))
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
