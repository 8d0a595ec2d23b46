"""Action-graph -> per-frame object vectors and boxes (reference: models/graph_models/model.py:23-180)."""
import torch
import torch.nn as nn

from ..attribute_embed import AttributeEmbeddings
from ..layers import Embedding, FusedMLP, Linear, build_mlp
from .graph import GraphTripleConv


def decode_actions(actions, timesteps, pad_action):
    """Clock-edge decoding of the action rows [s, a, o, t1, t2, x_end, y_end] (model.py:108-122): relative time of every
    frame inside every action window; actions outside their window become __padding__."""
    ar = actions.unsqueeze(1).expand(actions.size(0), timesteps, actions.size(1), actions.size(2))
    sa, a, oa, f1, f2, x_end, y_end = [x.squeeze(-1) for x in ar.chunk(7, dim=-1)]
    t = torch.arange(timesteps, dtype=torch.float32, device=actions.device).view(1, timesteps, 1)
    f1, f2 = f1.float(), f2.float()
    rel = (t / timesteps) * (f2 - f1 + 1e-6) + f1
    included = (rel >= 0) * (rel <= 1)
    # (torch.where, not a[~included] = pad: boolean-mask assignment goes through nonzero(), a host synchronisation at the top of
    # every graph-net forward -- the GPU queue ran dry there and the ~2 000 launches of the graph nets were issue-bound)
    a = torch.where(included, a, torch.full_like(a, pad_action))
    temporal_triplets = torch.stack([sa, a, oa], dim=-1).long()
    return temporal_triplets, rel, x_end, y_end


def action_edge_vectors(acts_embeddings, a, x_end_t, y_end_t, r_t):
    """Action embeddings whose last three columns are overwritten by (x_end, y_end, relative time) (model.py:143-146);
    written as a concatenation so the three columns of the table receive no gradient, as with the in-place write.
    Any leading dimensions ([B, A] for one time step, [B, F, A] for all of them)."""
    av = acts_embeddings(a.long())
    return torch.cat([av[..., :-3], x_end_t.unsqueeze(-1), y_end_t.unsqueeze(-1), r_t.unsqueeze(-1)], dim=-1)


class Acts2LayoutModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        args = vars(opt)
        self.vocab = args["vocab"]
        self.image_size = args["image_size"]
        self.args = args
        emb = args["embedding_dim"]
        self.attribute_embedding = AttributeEmbeddings(self.vocab['attributes'], emb)
        self.pred_embeddings = Embedding(len(self.vocab['pred_idx_to_name']), emb)
        self.acts_embeddings = Embedding(len(self.vocab['action_idx_to_name']), emb)
        num_attributes = len(self.vocab['attributes'].keys())
        obj_input_dim = num_attributes * emb
        first = dict(obj_input_dim=obj_input_dim, object_output_dim=args["gconv_dim"], predicate_input_dim=emb,
                     predicate_output_dim=args["gconv_dim"], hidden_dim=args["gconv_hidden_dim"],
                     num_attributes=num_attributes, mlp_normalization=args["mlp_normalization"],
                     pooling=args["gconv_pooling"], loc_dim=4)
        general = dict(first, obj_input_dim=first["object_output_dim"], predicate_input_dim=args["gconv_dim"])
        self.gconvs = nn.ModuleList([GraphTripleConv(**cfg) for cfg in [first] + [general] * (args["gconv_num_layers"] - 1)])
        self.box_net = build_mlp([general["object_output_dim"], args["gconv_hidden_dim"], 4],
                                 batch_norm=args["mlp_normalization"], final_nonlinearity=None)
        self.obj_vecs_net = FusedMLP(Linear(obj_input_dim + 4, obj_input_dim, bias=False), nn.ReLU(inplace=True),
                                     Linear(obj_input_dim, obj_input_dim, bias=False), nn.ReLU())
        if args.get("mask_size"):
            raise NotImplementedError("mask_size > 0 (mask_net) is not on the training hot path (data/args.py:45)")

    def actions_data(self, triplets, actions):
        """The fourth return value of forward() alone: [triplets, temporal_triplets, relative_timesteps, locs] depends on the
        inputs only (no parameter), so a caller can have it before the recurrence has run."""
        temporal_triplets, rel, x_end, y_end = decode_actions(actions, triplets.size(1),
                                                              self.vocab["action_name_to_idx"]["__padding__"])
        return [triplets, temporal_triplets, rel, torch.stack([x_end, y_end], dim=-1)]

    def forward(self, objs, triplets, actions, boxes_gt=None, test_mode=False):
        """objs [B,O,A] i64, triplets [B,F,T,3] i64, actions [B,A,7] f32, boxes_gt [B,F,O,4] ->
        (temporal_obj_vecs [B,F,O,gconv_dim], boxes_pred [B,F,O,4], [triplets, temporal_triplets, relative_timesteps, locs])"""
        vocab = self.vocab
        timesteps = triplets.size(1)
        temporal_triplets, rel, x_end, y_end = decode_actions(actions, timesteps, vocab["action_name_to_idx"]["__padding__"])
        boxes_pred = [boxes_gt[:, 0]]
        emb = self.attribute_embedding(objs)
        tvecs = [torch.zeros(objs.size(0), objs.size(1), self.args["embedding_dim"], device=emb.device, dtype=emb.dtype)]
        only_temporal = self.args.get("only_temporal", False)
        # Everything of a time step that does not depend on the recurrence (the boxes) -- edge lists, validity masks, predicate /
        # action embeddings with their (x_end, y_end, t) columns -- is built for ALL steps at once, frame-major, so that the loop
        # below only slices: ~10 small launches per step, net and pass become ~10 per net and pass.
        s, a, o = temporal_triplets[..., 0], temporal_triplets[..., 1], temporal_triplets[..., 2]          # [B, F, A]
        a_edges = torch.stack([s, o], dim=-1)
        a_ind = a != vocab["action_name_to_idx"]["__padding__"]
        av = action_edge_vectors(self.acts_embeddings, a, x_end, y_end, rel)                                 # [B, F, A, D]
        if not only_temporal:
            ss, p, so = triplets[..., 0], triplets[..., 1], triplets[..., 2]                                 # [B, F, T]
            ind_all = torch.cat([p != vocab["pred_name_to_idx"]["__padding__"], a_ind], dim=2)
            edges_all = torch.cat([torch.stack([ss, so], dim=-1), a_edges.long()], dim=2)
            pv_all = torch.cat([self.pred_embeddings(p), av], dim=2)
        else:
            ind_all, edges_all, pv_all = a_ind, a_edges, av
        ind_all = ind_all.transpose(0, 1).contiguous()            # [F, B, ...]: a step is a contiguous slice
        edges_all = edges_all.transpose(0, 1).contiguous()
        pv_fm = pv_all.transpose(0, 1).contiguous()                     # [F, B, T, D]
        from ... import graph_program
        if emb.is_cuda and timesteps > 1 and graph_program.supported(self.args, T=pv_fm.shape[2]):
            # the whole recurrence as one persistent-kernel program per pass (ag2video_amd/graph_program.py)
            lin = [m for m in self.obj_vecs_net.children() if isinstance(m, Linear)]
            blin = [m for m in self.box_net.children() if isinstance(m, Linear)]
            ov, boxes = graph_program.run_recurrence('emb', emb, pv_fm[1:], edges_all[1:], ind_all[1:], boxes_gt[:, 0], lin,
                                                     list(self.gconvs), blin)
            tvecs = torch.cat([tvecs[0].unsqueeze(1), ov.permute(1, 0, 2, 3)], dim=1)
            boxes_pred = torch.cat([boxes_gt[:, :1], boxes.permute(1, 0, 2, 3)], dim=1)
            locs = torch.stack([x_end, y_end], dim=-1)
            return tvecs, boxes_pred, [triplets, temporal_triplets, rel, locs]
        pv_steps = pv_fm.unbind(0)        # (unbind: ONE stacked gradient in the backward pass)
        for t in range(1, timesteps):
            boxes_c = boxes_pred[-1]
            ov = self.obj_vecs_net(torch.cat([emb, boxes_c], dim=-1))
            ind, edges, pred_vecs = ind_all[t], edges_all[t], pv_steps[t]
            for gconv in self.gconvs:
                ov, pred_vecs = gconv(ov, pred_vecs, edges, ind)
            tvecs.append(ov)
            boxes_pred.append(boxes_c + self.box_net(ov))
        boxes_pred = torch.stack(boxes_pred).permute(1, 0, 2, 3)
        tvecs = torch.stack(tvecs, dim=1)
        locs = torch.stack([x_end, y_end], dim=-1)
        return tvecs, boxes_pred, [triplets, temporal_triplets, rel, locs]


models = {'graph': Acts2LayoutModel}
