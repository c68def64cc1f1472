"""MultiRefRestorationModel: the caller of the hot path (mirror of
basicsr/models/multi_ref_restoration_model.py:20-386 for what the shipped yml exercises).

Same option keys (network_g / network_map / network_extractor / path.* / train.*), same methods
(feed_data :190-195, optimize_parameters :197-279, test :281-294, get_current_log,
update_learning_rate, save / load of ``{'params': state_dict}`` checkpoints), same optimiser
layout: Adam with four parameter groups chosen by name (:60-89) --
    'offset' & 'small'  -> lr_relu3_offset     'offset' & 'medium' -> lr_relu2_offset
    other 'offset'      -> lr_offset           everything else     -> lr_g
Only the pixel (L1) branch of the loss zoo is implemented: it is the only one the shipped config
enables (yml:77-78); enabling another raises NotImplementedError instead of silently skipping it.

What differs underneath (SURVEY 7 "hard parts"):
  * the K references run as one k-major batch through extractor / matching / VGG19 / net_g;
  * net_extractor and net_map are frozen replicas under no_grad and are NOT wrapped in DDP (no
    gradient can reach them past the arg-max; wrapping them only adds a "unused parameter"
    failure mode); net_g is wrapped in DistributedDataParallel (RCCL all-reduce of its 94.8 MB);
  * `l.item()` logging is deferred: log_dict holds device scalars until get_current_log().
"""
import logging
import os
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch.nn.parallel import DistributedDataParallel

from ..archs import build_network
from ..utils.registry import MODEL_REGISTRY


class _MultiStepRestartLR(torch.optim.lr_scheduler._LRScheduler):
    """basicsr/models/lr_scheduler.py:6-33 (MultiStepLR with optional restarts)"""

    def __init__(self, optimizer, milestones, gamma=0.1, restarts=(0, ), restart_weights=(1, ), last_epoch=-1):
        from collections import Counter
        self.milestones, self.gamma = Counter(milestones), gamma
        self.restarts, self.restart_weights = list(restarts), list(restart_weights)
        assert len(self.restarts) == len(self.restart_weights), 'restarts and their weights do not match.'
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        if self.last_epoch in self.restarts:
            weight = self.restart_weights[self.restarts.index(self.last_epoch)]
            return [group['initial_lr'] * weight for group in self.optimizer.param_groups]
        if self.last_epoch not in self.milestones:
            return [group['lr'] for group in self.optimizer.param_groups]
        return [group['lr'] * self.gamma**self.milestones[self.last_epoch] for group in self.optimizer.param_groups]


@MODEL_REGISTRY.register()
class MultiRefRestorationModel:

    def __init__(self, opt):
        self.opt = opt
        if opt.get('num_gpu', 1) == 0:
            raise NotImplementedError('mrefsr_amd has no CPU path: num_gpu must be >= 1')
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.is_train = opt['is_train']
        self.schedulers, self.optimizers = [], []
        logger = logging.getLogger('basicsr')

        self.net_map = build_network(opt['network_map']).to(self.device).eval()
        self.net_extractor = build_network(opt['network_extractor']).to(self.device).eval()
        for net in (self.net_map, self.net_extractor):
            for p in net.parameters():
                p.requires_grad_(False)
        path = opt.get('path', {})
        if path.get('pretrain_network_feature_extractor'):
            self.load_network(self.net_extractor, path['pretrain_network_feature_extractor'], path.get('strict_load', True))

        self.net_g = build_network(opt['network_g']).to(self.device)
        if path.get('pretrain_network_g'):
            self.load_network(self.net_g, path['pretrain_network_g'], path.get('strict_load', True))
        if opt.get('dist', False):
            self.net_g = DistributedDataParallel(self.net_g, device_ids=[self.device.index],
                                                 find_unused_parameters=opt.get('find_unused_parameters', False),
                                                 broadcast_buffers=opt.get('broadcast_buffers', True))
        self.log_dict = OrderedDict()
        if self.is_train:
            self.net_g.train()
            train_opt = opt['train']
            groups = {'g': [], 'offset': [], 'relu3': [], 'relu2': []}
            for name, v in self.get_bare_model(self.net_g).named_parameters():
                if not v.requires_grad:
                    continue
                if 'offset' in name:
                    if 'small' in name:
                        logger.info(name)
                        groups['relu3'].append(v)
                    elif 'medium' in name:
                        logger.info(name)
                        groups['relu2'].append(v)
                    else:
                        groups['offset'].append(v)
                else:
                    groups['g'].append(v)
            self.optimizer_g = torch.optim.Adam(
                [{'params': groups['g']},
                 {'params': groups['offset'], 'lr': train_opt['lr_offset']},
                 {'params': groups['relu3'], 'lr': train_opt['lr_relu3_offset']},
                 {'params': groups['relu2'], 'lr': train_opt['lr_relu2_offset']}],
                lr=train_opt['lr_g'], weight_decay=train_opt.get('weight_decay_g', 0), betas=train_opt['beta_g'],
                capturable=self._train_graph_wanted(),   # step counters on the device: the update can be part of a hipGraph
                # torch's fused multi-tensor Adam: the same update (ref :90-104 builds a plain torch.optim.Adam) in ~13 launches
                # instead of ~60; train.fused_adam: false keeps the per-operation foreach form (under hipGraph replay the foreach
                # form with device-side step counters costs 14 ms per step: 52.6 against 37.2 ms)
                fused=bool(train_opt.get('fused_adam', True)) and self.device.type == 'cuda')
            self.optimizers.append(self.optimizer_g)
            self.init_training_settings()

    # ------------------------------------------------------------------ set-up
    def init_training_settings(self):
        train_opt = self.opt['train']
        for key in ('perceptual_opt', 'style_opt', 'texture_opt', 'gan_type'):
            if train_opt.get(key):
                raise NotImplementedError(f'train.{key}: only the L1 pixel loss of the shipped config is implemented')
        if self.opt.get('network_d'):
            raise NotImplementedError('network_d: the shipped config trains without a discriminator')
        if train_opt['pixel_weight'] > 0:
            if train_opt['pixel_criterion'] != 'L1Loss':
                raise NotImplementedError(f"pixel_criterion {train_opt['pixel_criterion']}: only L1Loss is implemented")
            self.pixel_weight = float(train_opt['pixel_weight'])
        else:
            self.pixel_weight = None
        self.net_g_pretrain_steps = train_opt['net_g_pretrain_steps']
        self.net_d_steps = train_opt.get('net_d_steps', 1)
        self.net_d_init_steps = train_opt.get('net_d_init_steps', 0)
        sched = dict(train_opt['scheduler'])
        stype = sched.pop('type')
        if stype not in ('MultiStepLR', 'MultiStepRestartLR'):
            raise NotImplementedError(f'Scheduler {stype} is not implemented yet.')
        for optimizer in self.optimizers:
            self.schedulers.append(_MultiStepRestartLR(optimizer, **sched))

    @staticmethod
    def get_bare_model(net):
        return net.module if isinstance(net, DistributedDataParallel) else net

    # ------------------------------------------------------------------ data
    def feed_data(self, data):
        """data: img_in_lq (B,3,h,w), img_in_up (B,3,4h,4w), img_ref_list (B,K,3,4h,4w), img_in (B,3,4h,4w)
        (the dict of multi_ref_dataset.py:127-134)."""
        self.img_in_lq = data['img_in_lq'].to(self.device, non_blocking=True)
        refs = data['img_ref_list'].to(self.device, non_blocking=True)
        self.num_refs = refs.shape[1]
        # k-major stack [K*B,3,H,W]; the reference's list(torch.unbind(dim=1)) is its K slices
        self.img_ref_stack = refs.transpose(0, 1).reshape(-1, *refs.shape[2:]).contiguous()
        self.img_ref_list = list(self.img_ref_stack.view(self.num_refs, -1, *refs.shape[2:]).unbind(0))
        if 'img_in' in data:
            self.gt = data['img_in'].to(self.device, non_blocking=True)
        self.match_img_in = data['img_in_up'].to(self.device, non_blocking=True)

    # ------------------------------------------------------------------ the hot path
    def _forward(self):
        k = self.num_refs
        from .. import hip
        hip.amax_pool_reset()   # the max |out| words of this pass's launches (archs/nhwc.py: Winograd input scales): zeroed, slot 0
        with torch.no_grad():
            f1, f2 = self.net_extractor.forward_stacked(self.match_img_in, self.img_ref_stack)
            pre_offset, self.max_idx = self.net_map.offsets(f1, f2)
            img_ref_feat = self.net_map.vgg(self.img_ref_stack)
        return self.net_g(self.img_in_lq, pre_offset, img_ref_feat, k=k)

    range_fallbacks = 0   # batches re-run on the range-free kernels because an activation left the fp16 range

    def _range_tripped(self, what):
        """the default kernels split fp32 operands into fp16 pairs (|activation| < 65504) and raise a device flag
        otherwise: one 4-byte readback per batch; True -> the caller re-runs the batch under hip.range_free()"""
        from .. import hip
        if not hip.conv_range_tripped():
            return False
        self.range_fallbacks += 1
        logging.getLogger('basicsr').warning(
            f'{what}: an activation left the fp16 range of the split kernels; batch re-run on the bf16 three-term kernels '
            f'(no range limit, ~1.5x slower); {self.range_fallbacks} such batch(es) so far')
        return True

    def _loss_and_backward(self, step):
        """L1 branch of ref :197-279: returns True when a gradient was produced (the optimiser may step)"""
        if step <= self.net_g_pretrain_steps:
            l_pix = self.pixel_weight * F.l1_loss(self.output, self.gt)
            l_pix.backward()
            self.log_dict['l_pix'] = l_pix.detach()
            return True
        if (step - self.net_g_pretrain_steps) % self.net_d_steps == 0 and \
                (step - self.net_g_pretrain_steps) > self.net_d_init_steps:
            l_g_total = 0
            if self.pixel_weight is not None:
                l_g_pix = self.pixel_weight * F.l1_loss(self.output, self.gt)
                l_g_total = l_g_total + l_g_pix
                self.log_dict['l_g_pix'] = l_g_pix.detach()
            l_g_total.backward()
            return True
        return False

    # ------------------------------------------------------------------ hipGraph replay of the training step
    def _train_graph_wanted(self):
        """opt['train']['hip_graph'] or MREFSR_TRAIN_GRAPH=1 (EXPERIMENTAL, off by default: 1-2 % at the shipped patch size; see the
        fence at the end of _optimize_graphed): forward + backward are captured once per input shape and replayed; the Adam update is
        a second graph, replayed after the fp16-range flag has been read.  Single process only (a DDP all-reduce is not captured)."""
        return (bool((self.opt.get('train') or {}).get('hip_graph')) or os.environ.get('MREFSR_TRAIN_GRAPH', '0') == '1') \
            and not self.opt.get('dist', False)

    _TRAIN_INPUTS = ('img_in_lq', 'match_img_in', 'img_ref_stack', 'gt')
    _GRAPH_WARMUP = 3   # eager steps per input shape before capture (lazy kernel attributes, workspaces, MIOpen find results)
    # Experiment knobs of tools/train_graph_replay_fault.py (set on the class by the reproducer's child process, never by the product):
    #   _REPLAY_FENCE    False: no host fence behind the update graph (the fault shows) | 'sleep' | 'event' | 'device': other waits in its place
    #   _GRAPH_VARIANT   'shared_pool' (shipped: the update graph allocates from the forward / backward graph's pool) | 'own_pool' |
    #                    'one_graph' (forward, backward and update captured as ONE executable) | 'pack_outside' (the weight-pack launch
    #                    of begin_step() runs eagerly in front of each replay instead of inside the graph)
    _REPLAY_FENCE = True
    _GRAPH_VARIANT = 'shared_pool'

    def _optimize_graphed(self, step):
        """True when the step was taken by graph replay"""
        from .. import hip
        from ..archs import nhwc_train
        if step <= self.net_g_pretrain_steps or self.net_d_steps != 1 or step <= self.net_g_pretrain_steps + self.net_d_init_steps:
            return False   # (the phases of ref :197-279 differ in what they run and log: only the steady one is captured)
        # (the learning rates are NOT part of the key: the captured update reads them from device tensors, refreshed below when a
        # scheduler has moved them -- update_learning_rate with warm-up, base_model.py:172-193, moves them every iteration)
        key = (tuple(tuple(getattr(self, n).shape) for n in self._TRAIN_INPUTS), self.num_refs, hip.packed_epoch(), nhwc_train.scale_epoch())
        st = self.__dict__.setdefault('_tgraph', {'key': None})
        if st['key'] != key:
            changes = st.get('changes', 0) + (st['key'] is not None)
            st.clear()
            st.update(key=key, eager=0, fb=None, changes=changes)
            hip.release_capture_workspaces()
            if changes == 8:   # e.g. a scheduler that moves the learning rate every iteration: the key never settles
                logging.getLogger('basicsr').warning(
                    'hip_graph (training): the capture key (input shapes, weight scales) keeps changing; steps run eagerly.')
        if st['fb'] is None:
            if st['eager'] < self._GRAPH_WARMUP:
                st['eager'] += 1
                return False
            static = {n: getattr(self, n).clone() for n in self._TRAIN_INPUTS}
            for n, t in static.items():
                setattr(self, n, t)
            dyn = [m for m in self.net_g.modules() if hasattr(m, '_offset_count')]
            before = [m._offset_count for m in dyn]
            # nothing may keep the eager steps' autograd graphs (and their AccumulateGrad nodes, bound to the eager stream) alive
            self.output = None
            self.optimizer_g.zero_grad(set_to_none=True)
            import gc
            gc.collect()
            logging.getLogger('basicsr').warning(
                'hip_graph (training) is EXPERIMENTAL: hipStreamSynchronize is called behind every replayed update (without it the runtime of '
                'ROCm 7.2 faults after 25-50 graph launches: profiles/r5_train_graph_replay_fault.txt)')
            variant = self._GRAPH_VARIANT
            fb, upd = torch.cuda.CUDAGraph(), (None if variant == 'one_graph' else torch.cuda.CUDAGraph())
            lr_val = [float(pg['lr']) for pg in self.optimizer_g.param_groups]
            lr_dev = [torch.tensor(v, device=self.device, dtype=torch.float32) for v in lr_val]
            if variant == 'pack_outside':
                nhwc_train.begin_step()
            try:
                with torch.cuda.graph(fb):
                    if variant != 'pack_outside':
                        nhwc_train.begin_step()   # the packed weight copies are refreshed by the graph itself (one launch)
                    self.output = self._forward()
                    self._loss_and_backward(step)
                    if upd is None:
                        for pg, t in zip(self.optimizer_g.param_groups, lr_dev):
                            pg['lr'] = t
                        self.optimizer_g.step()
                if upd is not None:
                    for pg, t in zip(self.optimizer_g.param_groups, lr_dev):
                        pg['lr'] = t           # the captured update reads its learning rates from device memory ...
                    with (torch.cuda.graph(upd) if variant == 'own_pool' else torch.cuda.graph(upd, pool=fb.pool())):
                        self.optimizer_g.step()
            finally:
                for pg, v in zip(self.optimizer_g.param_groups, lr_val):
                    pg['lr'] = v       # ... while the schedulers keep working on plain numbers
            keep, ptrs = nhwc_train.capture_state()
            # the graphs have baked in addresses of eagerly allocated buffers that global caches own (pack job table, packed copies,
            # workspaces, zero chunks): they live as long as the graphs, and a replay is refused once any of them has moved
            st.update(fb=fb, upd=upd, static=static, out=self.output, idx=self.max_idx, log=dict(self.log_dict), dyn=dyn,
                      counts=[m._offset_count - b for m, b in zip(dyn, before)], keep=(keep, hip.capture_refs()), ptrs=ptrs, lr_val=lr_val, lr_dev=lr_dev)
        else:
            if nhwc_train.capture_state()[1] != st['ptrs']:   # (an eager pass in between rebuilt the pack table / a parameter moved)
                logging.getLogger('basicsr').warning('hip_graph (training): a buffer the captured step reads has moved; recapturing')
                st.clear()
                st.update(key=None)
                return False
            for n in self._TRAIN_INPUTS:
                st['static'][n].copy_(getattr(self, n))
                setattr(self, n, st['static'][n])
            for m, c in zip(st['dyn'], st['counts']):
                m._offset_count += c
        for i, pg in enumerate(self.optimizer_g.param_groups):   # a scheduler has moved a learning rate: one 4-byte fill, the graphs stay
            v = float(pg['lr'])
            if v != st['lr_val'][i]:
                st['lr_dev'][i].fill_(v)
                st['lr_val'][i] = v
        if self._GRAPH_VARIANT == 'pack_outside':
            nhwc_train.begin_step()
        st['fb'].replay()
        self.output, self.max_idx = st['out'], st['idx']
        self.log_dict.update(st['log'])
        if self._range_tripped('optimize_parameters'):   # rare: redo this step eagerly on the range-free kernels
            nhwc_train.reset_scales()
            self.optimizer_g.zero_grad()
            with hip.range_free():
                self.output = self._forward()
                stepped = self._loss_and_backward(step)
            hip.conv_range_tripped()
            if stepped:
                self.optimizer_g.step()
            return True
        if st['upd'] is not None:
            st['upd'].replay()
        # hipStreamSynchronize behind the update graph.  Without it a run of replayed steps ends in a GPU memory access fault after
        # 25-50 replays (ROCm 7.2; never in eager mode).  It is not an ordering fence: an event recorded here and waited for on the
        # host (the GPU provably idle at the next launch), or a 20-ms sleep, do NOT remove the fault, hipStreamSynchronize /
        # hipDeviceSynchronize do; the update graph in a pool of its own, or forward + backward + update as ONE executable, fault
        # alike; every fault address is page 0x37 / 0x38 of a 2-MB block (220 KB = the kernel-argument segments of one launch of this
        # ~700-node graph).  What the call does is make the runtime retire its per-launch bookkeeping of completed graph launches
        # (profiles/r5_train_graph_replay_fault.txt).  Cost: ~0.2 ms of host work not overlapped.
        if self._REPLAY_FENCE is True:
            torch.cuda.current_stream().synchronize()
        elif self._REPLAY_FENCE == 'sleep':      # (experiments: a host wait without any HIP call)
            import time
            time.sleep(0.02)
        elif self._REPLAY_FENCE == 'event':      # (experiments: an event recorded behind the update graph, waited for on the host)
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()
        elif self._REPLAY_FENCE == 'device':     # (experiments: hipDeviceSynchronize instead of the stream's)
            torch.cuda.synchronize()
        return True

    def optimize_parameters(self, step):
        from .. import hip
        from ..archs import nhwc_train
        nhwc_train.check_scales()   # cached fp16 weight scales of the training convolutions still valid? (device side)
        if self._train_graph_wanted() and self._optimize_graphed(step):
            return
        self.optimizer_g.zero_grad()
        nhwc_train.begin_step()     # every packed copy of net_g's weights refreshed in one launch (they changed in optimizer_g.step())
        self.output = self._forward()
        stepped = self._loss_and_backward(step)
        if self._range_tripped('optimize_parameters'):   # the frozen feature networks and the DCN forward run on the split kernels
            nhwc_train.reset_scales()
            self.optimizer_g.zero_grad()
            with hip.range_free():
                self.output = self._forward()
                stepped = self._loss_and_backward(step)
            hip.conv_range_tripped()
        if stepped:
            self.optimizer_g.step()

    def test(self):
        from .. import hip
        self.net_g.eval()
        with torch.no_grad():
            hip.verify_packed(self.device)   # packed weight copies still match their parameters? (one launch, read with the range flag)
            self.output = self._forward_graphed() if self._use_graph() else self._forward()
            tripped = self._range_tripped('test')
            if hip.packed_stale():   # a parameter was edited through .data: drop every packed copy and repeat the pass
                logging.getLogger('basicsr').warning('test: a parameter changed without a version bump (.data write?); packed weights rebuilt')
                hip.invalidate_packed()
                self.output = self._forward()
                tripped = self._range_tripped('test')
            if tripped:
                with hip.range_free():
                    self.output = self._forward()
                hip.conv_range_tripped()
        self.net_g.train()

    # ------------------------------------------------------------------ hipGraph replay of the inference pass
    _INPUTS = ('img_in_lq', 'match_img_in', 'img_ref_stack')

    def _use_graph(self):
        """opt['val']['hip_graph'] or MREFSR_GRAPH=1: the whole inference pass (240 launches) is captured once per
        input shape and parameter version into a hipGraph and replayed, taking ~50 us of host work per launch off
        the critical path.  Off by default: measured on MI355X it changes nothing, neither at the benchmark shape
        (GPU-bound) nor at LR 40x40 (12.5 ms per sample either way: there each convolution launch is bound by the
        serial K loop of a single block, ~60 us for a 256-channel layer, not by the host) -- it only helps when
        the host is the slower side."""
        return bool((self.opt.get('val') or {}).get('hip_graph')) or os.environ.get('MREFSR_GRAPH', '0') == '1'

    def _graph_key(self):
        from ..archs import nhwc
        from .. import hip
        versions = tuple((p._version, p.data_ptr()) for net in (self.net_g, self.net_extractor, self.net_map) for p in net.parameters())
        shapes = tuple(tuple(getattr(self, n).shape) for n in self._INPUTS)
        return shapes, self.num_refs, nhwc.TERMS, nhwc.BF16, hip.packed_epoch(), hash(versions)

    def _forward_graphed(self):
        key = self._graph_key()
        cache = self.__dict__.setdefault('_graphs', {})
        entry = cache.get(key)
        if entry is None:
            cache.clear()                              # one shape / parameter version at a time: graphs pin their buffers
            from .. import hip
            hip.release_capture_workspaces()
            static = {n: getattr(self, n).clone() for n in self._INPUTS}
            for n, t in static.items():
                setattr(self, n, t)
            dyn = [m for m in self.net_g.modules() if hasattr(m, '_offset_count')]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):              # eager warm-up: weight packing, workspaces, lazy kernel attributes
                self._forward()
            torch.cuda.current_stream().wait_stream(side)
            before = [m._offset_count for m in dyn]
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._forward()
            counts = [m._offset_count - b for m, b in zip(dyn, before)]
            entry = cache[key] = (graph, static, out, self.max_idx, dyn, counts)
            graph.replay()                             # (capture does not execute)
        else:
            graph, static, out, max_idx, dyn, counts = entry
            for n in self._INPUTS:
                static[n].copy_(getattr(self, n))
            graph.replay()
            for m, c in zip(dyn, counts):
                m._offset_count += c
        self.max_idx = entry[3].clone()
        return entry[2].clone()

    def check_numeric_range(self):
        """test() and optimize_parameters() already handle the fp16-split range flag themselves (re-run on the range-free
        kernels, counted in ``range_fallbacks``); this raises FloatingPointError if the flag is (still) set, for callers
        that launch parts of the path directly"""
        from .. import hip
        hip.check_conv_range()

    # ------------------------------------------------------------------ validation (ref :310-386)
    def validation(self, dataloader, current_iter, tb_logger, save_img=False):
        if self.opt.get('dist', False):
            return self.dist_validation(dataloader, current_iter, tb_logger, save_img)
        return self.nondist_validation(dataloader, current_iter, tb_logger, save_img)

    def dist_validation(self, dataloader, current_iter, tb_logger, save_img):
        if self.opt.get('rank', 0) == 0:  # rank 0 only, like the reference (:312-314)
            return self.nondist_validation(dataloader, current_iter, tb_logger, save_img)

    def nondist_validation(self, dataloader, current_iter, tb_logger, save_img):
        """per image: feed_data -> test -> uint8 image -> crop the dataset's zero padding -> [PNG] -> PSNR (RGB), PSNR (Y),
        SSIM (Y) with crop_border from the options (ref :324-386).  save_img writes
        path.visualization/<img>/<img>_<iter>.png while training, path.visualization/<dataset>/<img>_<name>[_suffix].png when testing."""
        from ..metrics import calculate_psnr, calculate_ssim, imwrite, tensor2img
        logger = logging.getLogger('basicsr')
        dataset_name = getattr(getattr(dataloader, 'dataset', None), 'opt', {}).get('name', 'val')
        psnrs, psnrs_y, ssims_y = [], [], []
        for idx, val_data in enumerate(dataloader):
            lq_path = val_data.get('lq_path', [f'{idx:04d}'])
            img_name = os.path.splitext(os.path.basename(lq_path[0] if isinstance(lq_path, (list, tuple)) else lq_path))[0]
            self.feed_data(val_data)
            self.test()
            sr_img = tensor2img(self.output[:1])
            gt_img = tensor2img(self.gt[:1])
            if 'padding' in val_data and val_data['padding']:
                oh, ow = [int(v) for v in val_data['original_size']][:2]
                sr_img, gt_img = sr_img[:oh, :ow], gt_img[:oh, :ow]
            if save_img:
                vis = self.opt['path']['visualization']
                if self.opt['is_train']:
                    save_path = os.path.join(vis, img_name, f'{img_name}_{current_iter}.png')
                else:
                    suffix = f"_{self.opt['suffix']}" if self.opt.get('suffix') else ''
                    save_path = os.path.join(vis, dataset_name, f"{img_name}_{self.opt['name']}{suffix}.png")
                imwrite(sr_img, save_path)
            cb = self.opt['crop_border']
            psnrs.append(calculate_psnr(sr_img, gt_img, crop_border=cb))
            psnrs_y.append(calculate_psnr(sr_img, gt_img, crop_border=cb, test_y_channel=True))
            ssims_y.append(calculate_ssim(sr_img, gt_img, crop_border=cb, test_y_channel=True))
            if not self.is_train:
                logger.info(f'# img {img_name} # PSNR: {psnrs[-1]:.4e} # PSNR_Y: {psnrs_y[-1]:.4e} # SSIM_Y: {ssims_y[-1]:.4e}.')
        n = max(len(psnrs), 1)
        avg_psnr, avg_psnr_y, avg_ssim_y = sum(psnrs) / n, sum(psnrs_y) / n, sum(ssims_y) / n
        logger.info(f'# Validation {dataset_name} # PSNR: {avg_psnr:.4e} # PSNR_Y: {avg_psnr_y:.4e} # SSIM_Y: {avg_ssim_y:.4e}.')
        if tb_logger:
            tb_logger.add_scalar('psnr', avg_psnr, current_iter)
            tb_logger.add_scalar('psnr_y', avg_psnr_y, current_iter)
            tb_logger.add_scalar('ssim_y', avg_ssim_y, current_iter)
        return dict(psnr=avg_psnr, psnr_y=avg_psnr_y, ssim_y=avg_ssim_y)

    def save(self, epoch, current_iter):
        """net_g checkpoint as {'params': state_dict} under path.models (ref :304-308, base_model.py:198-226)"""
        models_dir = self.opt.get('path', {}).get('models')
        if models_dir and self.opt.get('rank', 0) == 0:
            name = 'latest' if current_iter == -1 else current_iter
            self.save_network(self.net_g, os.path.join(models_dir, f'net_g_{name}.pth'))

    def save_training_state(self, epoch, current_iter):
        """optimizer / scheduler states as {epoch, iter, optimizers, schedulers} in
        path.training_states/<iter>.state (base_model.py:309-338); rank 0 only"""
        states_dir = self.opt.get('path', {}).get('training_states')
        if current_iter == -1 or not states_dir or self.opt.get('rank', 0) != 0:
            return
        state = {'epoch': epoch, 'iter': current_iter, 'optimizers': [o.state_dict() for o in self.optimizers],
                 'schedulers': [s.state_dict() for s in self.schedulers]}
        os.makedirs(states_dir, exist_ok=True)
        torch.save(state, os.path.join(states_dir, f'{current_iter}.state'))

    def resume_training(self, resume_state):
        """reload optimizers and schedulers from a save_training_state dict (base_model.py:343-356)"""
        resume_optimizers, resume_schedulers = resume_state['optimizers'], resume_state['schedulers']
        assert len(resume_optimizers) == len(self.optimizers), 'Wrong lengths of optimizers'
        assert len(resume_schedulers) == len(self.schedulers), 'Wrong lengths of schedulers'
        for o, state in zip(self.optimizers, resume_optimizers):
            o.load_state_dict(state)
        for sch, state in zip(self.schedulers, resume_schedulers):
            sch.load_state_dict(state)

    # ------------------------------------------------------------------ bookkeeping
    def get_current_log(self):
        return OrderedDict((k, v.item() if torch.is_tensor(v) else v) for k, v in self.log_dict.items())

    def get_current_visuals(self):
        out = OrderedDict(img_in_lq=self.img_in_lq.detach().cpu(), rlt=self.output.detach().cpu())
        if hasattr(self, 'gt'):
            out['gt'] = self.gt.detach().cpu()
        return out

    def offset_guards(self):
        """mean |learned offset| of the three DynAgg since the last call (ref :70-73 warning)."""
        dar = self.get_bare_model(self.net_g).dyn_agg_restore
        return {n: getattr(dar, n).offset_guard() for n in ('small_dyn_agg', 'medium_dyn_agg', 'large_dyn_agg')}

    def update_learning_rate(self, current_iter, warmup_iter=-1):
        if current_iter > 1:
            for scheduler in self.schedulers:
                scheduler.step()
        if current_iter < warmup_iter:
            for optimizer, scheduler in zip(self.optimizers, self.schedulers):
                for group in optimizer.param_groups:
                    group['lr'] = group['initial_lr'] / warmup_iter * current_iter

    def get_current_learning_rate(self):
        return [param_group['lr'] for param_group in self.optimizers[0].param_groups]

    def save_network(self, net, path, param_key='params'):
        state = OrderedDict((k[7:] if k.startswith('module.') else k, v.cpu())
                            for k, v in self.get_bare_model(net).state_dict().items())
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save({param_key: state}, path)

    def load_network(self, net, load_path, strict=True, param_key='params'):
        load_net = torch.load(load_path, map_location='cpu')
        if param_key is not None and param_key in load_net:
            load_net = load_net[param_key]
        load_net = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in load_net.items())
        self.get_bare_model(net).load_state_dict(load_net, strict=strict)


@MODEL_REGISTRY.register()
class RefRestorationModel(MultiRefRestorationModel):
    """Single-reference twin (basicsr/models/ref_restoration_model.py:20-375): `network_g` = RestorationNet,
    `network_extractor` = ContrasExtractorSep, data dict with one `img_ref` (B,3,4h,4w) (:190-194).  Same
    optimizer groups, schedulers, losses, validation and checkpoint layout as the multi-reference model."""
    _INPUTS = ('img_in_lq', 'match_img_in', 'img_ref')

    def feed_data(self, data):
        self.img_in_lq = data['img_in_lq'].to(self.device, non_blocking=True)
        self.img_ref = data['img_ref'].to(self.device, non_blocking=True)
        self.num_refs = 1
        self.img_ref_stack = self.img_ref
        self.img_ref_list = [self.img_ref]
        if 'img_in' in data:
            self.gt = data['img_in'].to(self.device, non_blocking=True)
        self.match_img_in = data['img_in_up'].to(self.device, non_blocking=True)

    def _forward(self):
        with torch.no_grad():   # frozen feature networks (:197-199, :278-280)
            features = self.net_extractor(self.match_img_in, self.img_ref)
            pre_offset, self.max_idx = self.net_map.offsets(features['dense_features1'], features['dense_features2'])
            img_ref_feat = self.net_map.vgg(self.img_ref)
        return self.net_g(self.img_in_lq, pre_offset, img_ref_feat)
