"""Scene/action-graph convolution (reference: models/graph_models/graph.py:16-107)."""
import torch.nn as nn

from ... import ops
from ..layers import Linear, build_mlp


def _init_weights(module):
    if isinstance(module, Linear):
        nn.init.kaiming_normal_(module.weight)


class GraphTripleConv(nn.Module):
    """One graph-convolution layer over (subject, predicate, object) triples with masked average pooling.

    The reference gathers / pools with a Python loop over the batch and boolean-mask indexing; here the whole batch is
    one gather launch, two fused GEMM+ReLU launches, one segmented-reduce pooling launch and two more GEMMs."""

    def __init__(self, obj_input_dim, object_output_dim, predicate_input_dim, predicate_output_dim, hidden_dim,
                 num_attributes, loc_dim=4, pooling='avg', mlp_normalization='none', return_new_p_vecs=True):
        super().__init__()
        assert pooling == 'avg', 'only avg pooling is on the hot path (gconv_pooling default, data/args.py:51)'
        self.return_new_p_vecs = return_new_p_vecs
        self.hidden_dim = hidden_dim
        self.num_attributes = num_attributes
        self.predicate_output_dim = predicate_output_dim
        self.pooling = pooling
        self.net1 = build_mlp([2 * obj_input_dim + predicate_input_dim, hidden_dim, 2 * hidden_dim + predicate_output_dim],
                              batch_norm=mlp_normalization, final_nonlinearity='relu')
        self.net1.apply(_init_weights)
        self.net2 = build_mlp([hidden_dim, hidden_dim, object_output_dim], batch_norm=mlp_normalization,
                              final_nonlinearity='relu')
        self.net2.apply(_init_weights)

    def forward(self, obj_vecs, pred_vecs, edges, pred_indicators):
        """obj_vecs [B,O,Din], pred_vecs [B,T,Dp], edges int64 [B,T,2], pred_indicators bool [B,T]."""
        O = obj_vecs.size(1)
        Hd, Dp = self.hidden_dim, self.predicate_output_dim
        cur_t = ops.triple_gather(obj_vecs, pred_vecs, edges)
        new_t = self.net1(cur_t)
        new_p = new_t[:, :, Hd:Hd + Dp]
        pooled = ops.triple_pool(new_t, edges, pred_indicators, O, Hd, Dp)
        new_obj = self.net2(pooled)
        if not self.return_new_p_vecs:
            new_p = pred_vecs
        return new_obj, new_p
