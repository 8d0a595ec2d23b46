"""Model facade with the reference's module surface (reference: models/meta_models.py:9-72).

``AG2VideoModel(opt, device)`` and ``MetaDiscriminatorModel(opt)`` expose the same attributes, forward signature,
return tuples and state_dict keys (492 / 56 keys at any width) as the reference, so scripts/train.py's loop drives them
unchanged; underneath, every layer runs on the gfx950 HIP kernels of ag2video_amd/csrc.
"""
import torch
import torch.nn as nn

from .graph_models.model import models
from .spade_models.networks.discriminator import MultiscaleActionDiscriminator, MultiscaleDiscriminatorT
from .spade_models.networks.generator import Layout2VidGenerator
from .spade_models.networks.sync_batchnorm import DataParallelWithCallback


class AG2VideoModel(nn.Module):
    def __init__(self, opt, device):
        super().__init__()
        self.args = vars(opt)
        self.vocab = self.args["vocab"]
        ids = self.args['gpu_ids']
        self.acts_to_boxes = DataParallelWithCallback(models[self.args["layout_arch"]](opt), device_ids=ids).to(device)
        if not self.args.get("coupled_motion_apperance"):
            objs_model = "graph" if self.args["layout_arch"] == 'gt_layout' else self.args["layout_arch"]
            self.acts_to_objs = DataParallelWithCallback(models[objs_model](opt), device_ids=ids).to(device)
        self.layout_to_video = DataParallelWithCallback(Layout2VidGenerator(opt), device_ids=ids).to(device)
        # engine.TrainStep sets a second HIP stream here: work whose result the rest of the forward pass does not read runs on it
        self.side_stream = self.side_stream_b = None

    def forward(self, imgs, objs, triplets, actions, boxes_gt=None, test_mode=False, use_gt=False, graph_only=False):
        """-> (imgs_pred, boxes_pred, flows_pred, conf_pred, actions_data), or boxes_pred when graph_only."""
        side = self.side_stream
        if side is not None and not graph_only and use_gt and not self.args.get("coupled_motion_apperance") and imgs.is_cuda:
            # Teacher forcing: the generator reads ground-truth boxes and the appearance net's vectors, so the box net's 15-step
            # recurrence (one persistent program, ~2.6 ms on <= 64 CUs) is only an OUTPUT of this call: it runs on its own stream beside
            # the appearance net's program; the caller's stream joins before this function returns.
            main = torch.cuda.current_stream()
            entry = main.record_event()
            # Round 6: the box net's program is launched FIRST, on its own stream, so that it runs beside the appearance net's (and the
            # discriminator's layout-term program, engine.gd_step) instead of behind the generator's forward, where it would wait for the
            # CUs the persistent convolutions hold.  Both side streams only wait for an event recorded at entry (the inputs).
            side_b = getattr(self, 'side_stream_b', None) or side
            side_b.wait_event(entry)
            with torch.cuda.stream(side_b):
                _, boxes_pred, _ = self.acts_to_boxes(objs, triplets, actions, boxes_gt, test_mode=test_mode)
                boxes_pred.record_stream(main)
            # The appearance net's FORWARD is on the critical path (the generator's first convolutions read its vectors), but it is
            # issued on the side stream as well: autograd runs a node's backward on the stream of its forward, so the net's 15-step
            # BACKWARD recurrence -- the tail of the generator's backward pass -- runs there, under the
            # discriminator phase that engine.gd_step queues next on the main stream (Adam(G) is applied after that phase).
            side.wait_event(entry)
            with torch.cuda.stream(side):
                obj_vecs, _, actions_data = self.acts_to_objs(objs, triplets, actions, boxes_gt, test_mode=test_mode)
                ready = side.record_event()
            # weights-only work of the generator (spectral norm's power iterations) goes to the main stream BEFORE it waits for the
            # appearance net: it runs beside the three graph-net programs instead of behind them
            self.layout_to_video.module.prefetch_spectral_scales(boxes_gt.size(1), test_mode=test_mode)
            main.wait_event(ready)
            for t_ in [obj_vecs] + [a for a in actions_data if torch.is_tensor(a)]:
                t_.record_stream(main)
            imgs_pred, flows_pred, conf_pred = self.layout_to_video(imgs, objs, obj_vecs, boxes_gt, test_mode=test_mode)
            main.wait_stream(side_b)
            return imgs_pred, boxes_pred, flows_pred, conf_pred, actions_data
        obj_vecs, boxes_pred, actions_data = self.acts_to_boxes(objs, triplets, actions, boxes_gt, test_mode=test_mode)
        if graph_only:
            return boxes_pred
        if not self.args.get("coupled_motion_apperance"):
            # default: motion (boxes) and appearance (object vectors) come from two separately parameterised nets
            obj_vecs, _, actions_data = self.acts_to_objs(objs, triplets, actions, boxes_gt, test_mode=test_mode)
        boxes_in = boxes_gt if use_gt else boxes_pred.detach()
        imgs_pred, flows_pred, conf_pred = self.layout_to_video(imgs, objs, obj_vecs, boxes_in, test_mode=test_mode)
        return imgs_pred, boxes_pred, flows_pred, conf_pred, actions_data


class MetaDiscriminatorModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.args = vars(opt)
        self.init_act_discriminator(opt)

    def init_act_discriminator(self, opt):
        self.img_discriminator = MultiscaleActionDiscriminator(opt)
        self.img_discriminator.float()
        self.img_discriminator.train()
        params = list(self.img_discriminator.parameters())
        if getattr(opt, 'temporal_D', 0):
            # --temporal_D 1: the reference's temporal discriminator classes (discriminator.py:16-87), which its own
            # MetaDiscriminatorModel never constructs (meta_models.py:66-67).  Same Adam as the image discriminator.
            self.temporal_discriminator = MultiscaleDiscriminatorT(opt)
            self.temporal_discriminator.float()
            self.temporal_discriminator.train()
            params += list(self.temporal_discriminator.parameters())
        self.optimizer_d_img = torch.optim.Adam(params, lr=opt.learning_rate, betas=(opt.beta1, 0.999))

    def d_parameters(self):
        """Everything optimizer_d_img owns: the image discriminator (+ the temporal one with --temporal_D 1)."""
        ps = list(self.img_discriminator.parameters())
        if hasattr(self, 'temporal_discriminator'):
            ps += list(self.temporal_discriminator.parameters())
        return ps
