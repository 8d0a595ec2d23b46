"""Per-attribute object embeddings (reference: models/attribute_embed.py:16-46)."""
import torch
import torch.nn as nn

from .layers import Embedding, Linear


class AttributeEmbeddings(nn.Module):
    """One embedding table per attribute, concatenated, then Linear(n*d, n*d) when there is more than one attribute."""

    def __init__(self, attributes, embedding_dim, use_attr_fc_gen=False):
        super().__init__()
        num_attr = len(attributes)
        if num_attr > 1 or use_attr_fc_gen:
            self.attribute_fc_gen = Linear(num_attr * embedding_dim, num_attr * embedding_dim)
        for i, name in enumerate(list(attributes)):
            self.add_module("att_emb_%d" % i, Embedding(max(attributes[name].values()) + 1, embedding_dim))
        self.num_attr = num_attr

    def forward(self, x):
        """x: [B, O, A] int64 -> [B, O, A*d]"""
        vecs = [self._modules["att_emb_%d" % k](x[:, :, k]) for k in range(x.size(-1))]
        v = torch.cat(vecs, dim=-1)
        if hasattr(self, 'attribute_fc_gen'):
            v = self.attribute_fc_gen(v)
        return v
