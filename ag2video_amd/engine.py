"""One training iteration of the reference loop (scripts/train.py:362-375,440-493) as a reusable object.

Builds the three pieces the reference's ``main`` builds -- ``AG2VideoModel``, ``MetaDiscriminatorModel`` (with its Adam)
and ``LossModel`` wrapped in ``DataParallelWithCallback`` -- plus the generator / graph optimizers with the reference's
parameter partition (train.py:365-368), and exposes the G step, D step and graph step in the reference's order.
scripts/train.py, bench.py, __graft_entry__.smoke() and the tests all drive this object.
"""
import os

import torch

from . import distributed as dist_utils
from . import ops
from .models.meta_models import AG2VideoModel, MetaDiscriminatorModel
from .models.spade_models.loss_model import LossModel
from .models.spade_models.networks.sync_batchnorm import DataParallelWithCallback
from .models.utils import batch_to


class TrainStep(object):
    def __init__(self, opt, device, weights_seed=None, adam_cls=None):
        """adam_cls: optimizer class override (tests only: a flat-buffer Adam whose step runs on the CPU, so that the gloo test drives
        the same two-phase gradient all-reduce as the GPU path); default FusedAdam on the HIP device."""
        self.opt = opt
        self.device = device
        # batch-norm statistics: the reference synchronises them over the devices of a DataParallel run and uses F.batch_norm on a single
        # device (sync_batchnorm/batchnorm.py:63-93); --sync_bn -1 (the default) follows it per world size
        sb = int(getattr(opt, 'sync_bn', -1))
        dist_utils.set_sync_bn(sb > 0 or (sb < 0 and dist_utils.is_dist() and dist_utils.world_size() > 1))
        if getattr(opt, 'lean_tape', 0):
            ops.set_lean_tape(True)
        self.model = AG2VideoModel(opt, device)
        self.model.float()
        self.discriminator = MetaDiscriminatorModel(opt)
        self.discriminator.to(device)
        self.gans_model = DataParallelWithCallback(LossModel(opt, discriminator=self.discriminator),
                                                   device_ids=opt.gpu_ids).to(device)
        if weights_seed is not None:
            from .weights import closed_form_fill_
            closed_form_fill_(self.model, seed=weights_seed, prefix='G.')
            closed_form_fill_(self.discriminator.img_discriminator, seed=weights_seed, prefix='D.')
            if hasattr(self.discriminator, 'temporal_discriminator'):
                closed_form_fill_(self.discriminator.temporal_discriminator, seed=weights_seed, prefix='T.')
            # (weights loaded through --vgg_weights are the objective: the closed-form filler only replaces a random init)
            if hasattr(self.gans_model.module, 'criterionVGG') and not getattr(self.gans_model.module, 'vgg_weights_loaded', False):
                closed_form_fill_(self.gans_model.module.criterionVGG, seed=weights_seed, prefix='V.')
        lr, b1 = opt.learning_rate, opt.beta1
        graph_params = list(self.model.acts_to_boxes.parameters())
        graph_ids = set(id(p) for p in graph_params)
        gen_params = [p for p in self.model.parameters() if id(p) not in graph_ids]   # train.py:367
        # Adam as in train.py:365-368 / meta_models.py:70-72; on the GPU the fused flat-buffer implementation (one kernel
        # per step).  (torch.optim.Adam is only used when the modules live on the CPU, i.e. in the host-logic tests.)
        if adam_cls is not None:
            Adam = adam_cls
        elif torch.device(device).type == 'cuda':
            from .optim import FusedAdam as Adam
        else:
            Adam = torch.optim.Adam
        self.optimizer_graph = Adam(graph_params, lr=lr, betas=(b1, 0.999))
        self.optimizer_generator = Adam(gen_params, lr=lr, betas=(b1, 0.999))
        # the discriminator's Adam was built before .to(device); rebuild it over the moved parameters
        self.discriminator.optimizer_d_img = Adam(self.discriminator.d_parameters(), lr=lr, betas=(b1, 0.999))
        self._gen_params, self._graph_params = gen_params, graph_params
        # bench.py (N > 1): HIP event pairs around the join of the generator's asynchronous all-reduce (= the time the compute stream
        # stalls for it: the EXPOSED part) and around the discriminator's synchronous one; None = not recorded
        self.comm_events = None
        self._d_params = self.discriminator.d_parameters()
        for m in (self.model, self.discriminator):
            m.train()
        # Second HIP stream for the launch-latency-bound graph nets whose results nothing on the main stream waits for: the box
        # net's forward recurrence (AG2VideoModel.forward) and the discriminator's layout terms, forward here and backward at
        # the tail of the discriminator's backward pass (autograd runs a node on the stream its forward ran on).
        self.side_stream = self.side_stream_b = self.side_stream_d = None
        # (deterministic mode keeps everything on one stream: two streams accumulating gradients interleave in a varying order)
        # Round 6: ON by default (AG2V_NO_SIDE_STREAM=1 turns it off).  The graph nets are persistent programs -- ONE launch each, 345
        # dependent operators behind grid barriers, ~2.6 ms during which a program uses <= 128 CUs and nothing else runs: measured
        # (tools/exp_graph_cost.py) 15.4 ms of the c3 step's wall time for the six launches.  The three FORWARD programs of a step (appearance
        # net, box net, the discriminator's layout terms) depend on the batch and on parameters only: launched together at the start of the
        # step on three streams with 64 workgroups each (graph_program.GRID) they run beside each other instead of one after the other.
        # Rounds 3-5 had the switch off: the per-operator graph nets of round 2 only ever ran in the gaps between convolutions (790.0 vs
        # 790.0 ms), and round 3's programs were queued BEHIND the generator's forward, where they wait for CUs the persistent
        # convolutions hold.
        if (torch.device(device).type == 'cuda' and os.environ.get('AG2V_NO_SIDE_STREAM') is None and not ops.is_deterministic()):
            # HIGH-priority streams: a program passes its grid barriers only once ALL its workgroups are resident, so when CUs free up the
            # dispatcher must hand them to a waiting program before it hands them to the thousands of workgroups a main-stream kernel has
            # queued -- a program that stays partly resident holds its CUs idle AND leaves the persistent convolutions (one workgroup per
            # CU) a second round of workgroups: seen as a single launch taking +20 ms in 2 of 6 bench runs before the priorities were set
            # (AG2V_SIDE_PRIORITY=0: the default priority, for A/B runs).
            prio = -1 if os.environ.get('AG2V_SIDE_PRIORITY', '-1') != '0' else 0
            self.side_stream = torch.cuda.Stream(device=device, priority=prio)
            self.side_stream_b = torch.cuda.Stream(device=device, priority=prio)
            self.side_stream_d = torch.cuda.Stream(device=device, priority=prio)
            self.model.side_stream = self.side_stream
            self.model.side_stream_b = self.side_stream_b
            from . import graph_program
            graph_program.GRID[0] = 64
            # (graph-net parameters of D receive gradients from nodes of both streams)
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        dist_utils.broadcast_parameters(list(self.model.parameters()) + self._d_params +
                                        list(self.model.buffers()) + list(self.discriminator.buffers()))

    def to_device(self, batch):
        return batch_to(batch, self.device)

    def check_health(self):
        """Raises RuntimeError when a graph-net program launched since the last call gave up at a grid barrier (results invalid).
        Costs a wait for 4-byte copies that are complete by any host read of the step; scripts/train.py and bench.py also call it
        once after their last step."""
        if torch.device(self.device).type == 'cuda':
            from . import graph_program
            graph_program.check_timeouts()

    def gd_step(self, batch, apply_updates=True, nan_guard=True):
        """Generator step then discriminator step on one clip batch (train.py:446-464, 522-525).
        nan_guard: the reference's NaN guard (train.py:450-453) -- a NaN in GAN_Img / GAN_Feat skips both updates of this
        iteration.  The decision is COLLECTIVE (distributed.any_rank): every rank skips or none does, so the gradient
        all-reduces below always pair up across ranks.  Returns (model_out, G_losses, D_losses); D_losses is None when skipped."""
        imgs, objs, boxes, triplets, actions, video_id = batch
        # (a trainer built outside deterministic mode and stepped inside it keeps this step on one stream: two streams accumulating
        # gradients interleave in a varying order)
        side_on = self.side_stream is not None and not ops.is_deterministic()
        self.model.side_stream = self.side_stream if side_on else None
        if side_on:
            adata = self.model.acts_to_boxes.module.actions_data(triplets, actions)      # (inputs only; a few small launches)
            step_start = torch.cuda.current_stream().record_event()
        if side_on:
            # the discriminator's layout terms (its graph-net program first) on their own stream, launched BEFORE the generator's forward
            # is queued: the program then runs beside the appearance net's and the box net's (AG2VideoModel.forward), ordered only
            # behind the start of this step
            self.gans_model.module.prefetch_layout_terms(batch, adata, self.side_stream_d, after=step_start)
        model_out = self.model(imgs, objs, triplets, actions, boxes_gt=boxes, test_mode=False, use_gt=True)
        G_losses = self.gans_model(batch, model_out, mode="compute_generator_loss")
        G_losses = {k: v.mean() for k, v in G_losses.items()}
        if nan_guard:
            watched = [G_losses[k] for k in ('GAN_Img', 'GAN_Feat') if k in G_losses]
            bad = bool(torch.isnan(torch.stack([w.reshape(()) for w in watched])).any().item())   # the one host read of the step
            # ... next to it: the timeout flags of every graph-net program run since the last check (this step's forward programs, the
            # previous step's backward programs) -- a run whose workgroups were not co-resident produced garbage and must not train on
            self.check_health()
            if dist_utils.any_rank(bad, self.device):
                if bad:
                    print("Error: video_id {} is NAN!".format(video_id))
                return model_out, G_losses, None
        self.optimizer_generator.zero_grad()
        # single process with a side stream: the appearance net's backward (side stream) keeps running under the discriminator phase;
        # this stream joins it before Adam(G) below.  With collectives the all-reduce launched next must see complete gradients.
        overlap_tail = side_on and not dist_utils.is_dist()
        with ops.accumulate_into_param_grads(join_streams=not overlap_tail):
            G_losses["total_loss"].backward()
        # One process per GPU: the generator's 500 MB gradient all-reduce is launched here and joined after the discriminator's
        # loss + backward (which read neither these gradients nor the generator's parameters: imgs_pred is detached), then
        # Adam(G) runs -- the same updates as the reference order (G step, then D loss), with the ring traffic behind ~150 ms of
        # convolutions.  Single process: nothing is pending and Adam(G) runs here, in the reference's order (unless the optional side
        # stream is on: then Adam(G) waits for the appearance net's backward behind the discriminator phase, `overlap_tail`).
        pending = dist_utils.average_gradients_begin(self._gen_params, flat=getattr(self.optimizer_generator, 'flat_grad', None))
        deferred = bool(pending.works) or overlap_tail      # Adam(G) after the discriminator phase (which reads neither G's gradients nor its parameters)
        if apply_updates and not deferred:
            self.optimizer_generator.step()
        D_losses = self.gans_model(batch, model_out, mode="compute_discriminator_loss")
        D_losses = {k: v.mean() for k, v in D_losses.items()}
        self.discriminator.optimizer_d_img.zero_grad()
        with ops.accumulate_into_param_grads():
            D_losses["total_img_loss"].backward()
        if side_on:      # the layout terms' backward ran there; their gradients are written by the ops themselves
            torch.cuda.current_stream().wait_stream(self.side_stream)
            torch.cuda.current_stream().wait_stream(self.side_stream_d)
        ops.join_direct_grad_streams()
        rec = self.comm_events is not None and dist_utils.is_dist() and torch.device(self.device).type == 'cuda'
        if deferred:
            if rec:
                e0 = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
            pending.finish()
            if rec:
                self.comm_events.append(('g_join', e0, torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))))
            if apply_updates:
                self.optimizer_generator.step()
        if rec:
            e0 = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
        dist_utils.average_gradients(self._d_params, flat=getattr(self.discriminator.optimizer_d_img, 'flat_grad', None))
        if rec:
            self.comm_events.append(('d_sync', e0, torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))))
        if apply_updates:
            self.discriminator.optimizer_d_img.step()
        return model_out, G_losses, D_losses

    def graph_step(self, batch, apply_updates=True):
        """Box-prediction step on a (longer) clip (train.py:483-493)."""
        imgs, objs, boxes, triplets, actions, video_id = batch
        boxes_pred = self.model(imgs, objs, triplets, actions, boxes_gt=boxes, test_mode=False, graph_only=True)
        L = self.gans_model(batch, boxes_pred, mode="compute_graph_loss")
        L = {k: v.mean() for k, v in L.items()}
        self.optimizer_graph.zero_grad()
        with ops.accumulate_into_param_grads():
            L["total_loss"].backward()
        dist_utils.average_gradients(self._graph_params, flat=getattr(self.optimizer_graph, 'flat_grad', None))
        if apply_updates:
            self.optimizer_graph.step()
        return boxes_pred, L
