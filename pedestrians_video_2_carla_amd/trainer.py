"""Minimal trainer: the Lightning loop the reference delegates to, reduced to what the train step needs
(reference modeling.py:275-282 ``pl.Trainer.from_argparse_args(...).fit(model, datamodule)``).

Hook order per step is Lightning's: ``on_train_batch_start`` -> ``training_step`` -> backward -> optimizer step.
Two execution modes:
  * eager: every launch issued from Python;
  * ``use_graph=True``: the step (model forward, HIP pose head forward/backward, optimizer) is captured once into a
    HIP graph on STATIC batch buffers owned by the trainer and replayed. Every new batch is *staged* first -- copied
    into the static buffers (ONE launch for all tensors, p2c_copy_group) and handed to ``flow.on_train_batch_start`` (per-batch constants:
    skeleton-type index, target-pair counts) -- outside the graph; passing the same batch object again (a resident
    batch, as bench.py does) stages nothing. With more than one rank the gradient all-reduce is captured into the
    step graph when every rank can do so, otherwise it stays outside (capture A: forward + backward; eager RCCL
    all-reduce; optimizer launch).
"""
import contextlib
import os
import sys
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from pedestrians_video_2_carla_amd.parallel.flat import FlatParameters, GradientExchange


def seed_everything(seed: int = 22742):
    """``pl.seed_everything(seed, workers=True)`` (reference modeling.py:350-351; default seed :120-121)."""
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ['PL_GLOBAL_SEED'] = str(seed)


def init_distributed(backend: Optional[str] = None) -> Dict[str, int]:
    """One process per GPU, rendezvous from the torchrun env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    forced = os.environ.get('P2C_FORCE_EXCHANGE', '0') == '1' and 'RANK' in os.environ   # one-rank rehearsal of the DP path
    if (world > 1 or forced) and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'      # "nccl" IS RCCL on ROCm
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return {'world_size': world, 'rank': rank, 'local_rank': local_rank}


def ranks_agree(ok: bool, device, checksum: Optional[torch.Tensor] = None, group=None) -> bool:
    """Eager agreement of all ranks on a local outcome: True only if EVERY rank passed ``ok`` and -- when given -- every
    rank holds the same finite ``checksum`` (a 0-dim float64 tensor). Two small all-reduces (MIN / MAX); every rank must
    call it the same number of times, whatever happened locally."""
    chk = checksum.detach().to(device=device, dtype=torch.float64).reshape(()) if (checksum is not None and ok) \
        else torch.zeros((), dtype=torch.float64, device=device)
    votes = torch.stack((torch.tensor(1.0 if ok else 0.0, dtype=torch.float64, device=device), chk))
    lo, hi = votes.clone(), votes.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(lo[0] > 0.5) and bool(torch.isfinite(lo[1])) and bool(torch.isfinite(hi[1])) and float(hi[1]) == float(lo[1])


class Trainer:
    def __init__(self, max_steps: int = 100, use_graph: bool = False, device: Optional[torch.device] = None,
                 flatten: bool = True, log_every_n_steps: int = 0, steps_per_epoch: Optional[int] = None):
        self.max_steps = max_steps
        self.use_graph = use_graph
        self.device = device
        self.flatten = flatten
        self.log_every_n_steps = log_every_n_steps
        self.steps_per_epoch = steps_per_epoch
        self.datamodule = None
        self.loggers: List = []
        self.global_step = 0
        self.current_epoch = 0
        self.flat: Optional[FlatParameters] = None
        self.exchange: Optional[GradientExchange] = None
        self.optimizers: List[torch.optim.Optimizer] = []
        self.lr_schedulers: List[dict] = []
        self._graphs = None
        self._static_loss = None
        self._static_batch = None
        self._staged_src = None
        self._group_copy = None
        self._direct = None
        self._unit = None
        self._packed = []
        self._opt_in_backward = False
        self._fused_seen = 0
        self._grad_sinks = False
        self._kept_graph = False
        self._graph_nodes = None
        self._handed_over = None

    # ------------------------------------------------------------------------------------------------------------
    def setup(self, flow, datamodule):
        self.datamodule = datamodule
        flow.trainer = self
        if self.device is not None:
            flow.to(self.device)
        flow.train()
        configs = flow.configure_optimizers()
        if self.flatten:
            self.flat = FlatParameters(flow.parameters())
            self.exchange = GradientExchange(self.flat)
            self.exchange.broadcast_parameters(0)
            self.exchange.broadcast_buffers(flow, 0)      # DDP broadcasts module buffers (BatchNorm statistics) at wrap time
            extra = {}
            if self.flat.flat_param.is_cuda:
                extra['fused'] = True              # one kernel for the whole (flat) parameter
                extra['capturable'] = bool(self.use_graph)
                # plugins with a fused backward write their parameter gradients straight into the flat buffer
                for m in flow.modules():
                    if hasattr(m, 'grad_sink'):
                        m.grad_sink = True
                self._grad_sinks = True          # K12 adds weight gradients straight into the flat gradient buffer's views
            if len(configs) != 1:
                raise ValueError('exactly one trainable plugin expected (ZeroTrajectory has no optimizer)')
            self.optimizers = [self.flat.rebuild_optimizer(configs[0]['optimizer'], **extra)]
            opt = self.optimizers[0]
            # modules with a re-laid-out copy of their weights (the fused MLP's LDS image) let the optimizer keep it current
            self._packed = [m for m in flow.modules() if hasattr(m, 'manage_packed_image')
                            and m.manage_packed_image(self.flat.flat_param, opt)]
            # single GPU: a module that owns every parameter can apply the optimizer step inside its own backward
            self._opt_in_backward = (not self.exchange.enabled and os.environ.get('P2C_FUSED_UPDATE', '1') == '1'
                                     and any(m.accept_fused_optimizer(opt, self.flat.flat_param) for m in self._packed
                                             if hasattr(m, 'accept_fused_optimizer')))
            self._fused_seen = getattr(opt, 'fused_steps_applied', 0)
            if hasattr(opt, 'grad_scale') and self.exchange.enabled:    # FlatAdamW folds the DP averaging into its pass
                opt.grad_scale = 1.0 / self.exchange.world
                self.exchange.average_here = False
        else:
            self.optimizers = [c['optimizer'] for c in configs]
        # LR schedulers (base_model.py:159-199) were built on the plugin's own optimizer: re-bind them to the one that steps
        self.lr_schedulers = []
        for c, opt in zip(configs, self.optimizers):
            sched = c.get('lr_scheduler')
            if sched is None:
                continue
            sched = dict(sched) if isinstance(sched, dict) else {'scheduler': sched}
            sched.setdefault('interval', 'epoch')        # Lightning's defaults
            sched.setdefault('frequency', 1)
            s = sched['scheduler']
            if s.optimizer is not opt:
                for g_new, g_old in zip(opt.param_groups, s.optimizer.param_groups):
                    g_new.setdefault('initial_lr', g_old.get('initial_lr', g_old['lr']))
                    g_new['lr'] = g_old['lr']
                s.optimizer = opt
            self.lr_schedulers.append(sched)
        return self

    def step_lr_schedulers(self, interval: str = 'epoch', monitor: Optional[Dict[str, float]] = None):
        """What Lightning does at the end of an epoch / step for the configured schedulers. ``monitor``: logged values for
        ReduceLROnPlateau (``{'val_loss/primary': ...}``); a plateau scheduler whose key is missing is skipped, as Lightning
        would raise only in strict mode. The captured optimizer launch reads the LR from device memory
        (FlatAdamW.sync_hyper), so the change reaches graph replays."""
        count = self.current_epoch if interval == 'epoch' else self.global_step
        for sched in self.lr_schedulers:
            if sched['interval'] != interval or count % sched['frequency'] != 0:
                continue
            s = sched['scheduler']
            if isinstance(s, torch.optim.lr_scheduler.ReduceLROnPlateau):
                value = (monitor or {}).get(sched.get('monitor'))
                if value is not None:
                    s.step(value)
            else:
                s.step()

    def _zero_grad(self):
        if self.flat is not None:
            if getattr(self.optimizers[0], 'zero_grad_in_step', False):
                return                  # the flat gradient starts zeroed and FlatAdamW leaves it zeroed after each step
            self.flat.zero_grad()
        else:
            for o in self.optimizers:
                o.zero_grad(set_to_none=True)

    def _forward_backward(self, flow, batch, batch_idx, batch_start: bool = True):
        self._zero_grad()
        # forward and backward happen back to back in here and nothing reads a loss VALUE in between (the reference's
        # per-loss isnan check is off unless strict_nan_check): the pose head may leave the last stage of its loss
        # reduction to the backward kernel, or run only in the backward at all (P2C_DEFER_FINALIZE=0|1|2, see
        # ops.deferred_loss_finalize)
        mode = int(os.environ.get('P2C_DEFER_FINALIZE', '2'))      # 2: the pose head runs once per step, in the backward
        defer = batch[0].is_cuda and not getattr(flow, 'strict_nan_check', False) and mode != 0
        from pedestrians_video_2_carla_amd import ops
        with (ops.deferred_loss_finalize(mode) if defer else contextlib.nullcontext()), ops.grad_sinks(self._grad_sinks):
            if batch_start:
                flow.on_train_batch_start(batch, batch_idx)
            out = flow.training_step(batch, batch_idx)
            loss = out['loss']
            if self._unit is None or self._unit.shape != loss.shape or self._unit.device != loss.device:
                self._unit = torch.ones_like(loss)   # root gradient, made once (backward() would fill one per step)
            loss.backward(gradient=self._unit)
        return loss.detach()

    def _optimizer_step(self):
        if self._opt_in_backward:
            # normally applied by the backward launch of the module that owns the parameters -- but only if that launch
            # really ran this step (module in eval(), host tensors, detached .grad views ... take other paths)
            applied = self.optimizers[0].fused_steps_applied
            if applied != self._fused_seen:
                self._fused_seen = applied
                return
        for o in self.optimizers:
            o.step()

    # ---- static batch (graph mode) -----------------------------------------------------------------------------------
    @staticmethod
    def _batch_tensors(batch):
        frames, targets, meta = batch
        out = [('frames', frames)]
        out += [('targets/' + k, v) for k, v in sorted(targets.items()) if isinstance(v, torch.Tensor)]
        out += [('meta/' + k, v) for k, v in sorted(meta.items()) if isinstance(v, torch.Tensor)]
        return out

    def _with_skel_type(self, batch):
        """meta['skel_type'] as a device tensor (data/carla/reference.py:skeleton_types_from_meta): inside a captured step
        the per-clip (age, gender) strings cannot be looked at again."""
        frames, targets, meta = batch
        st = meta.get('skel_type')
        if isinstance(st, torch.Tensor) and st.dtype == torch.int32 and st.device == frames.device:
            return batch
        # ALWAYS an int32 tensor on the frames' device, also for an int64 / host tensor and for the default adult-female
        # case without age / gender: the static batch then owns it and ProjectionModule.on_batch_start (whose `.to()` calls
        # are no-ops on it) hands the captured step that same storage for every staged batch
        from pedestrians_video_2_carla_amd.data.carla import reference as ref
        meta = dict(meta)
        meta['skel_type'] = ref.skeleton_types_from_meta(meta, batch_size=len(frames), strict=True, device=frames.device)
        return frames, targets, meta

    def stage_batch(self, flow, batch, batch_idx: int = 0):
        """Make ``batch`` the content of the static buffers the captured step reads (no-op for the batch OBJECT staged
        last: a resident batch), then run the flow's per-batch hook on them. Returns the static batch."""
        if self._static_batch is not None and batch is self._staged_src:
            return self._static_batch
        key, batch = batch, self._with_skel_type(batch)
        if self._static_batch is None:
            frames, targets, meta = batch
            clone = lambda d: {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}   # noqa: E731
            self._static_batch = (frames.clone(), clone(targets), clone(meta))
            named = self._batch_tensors(self._static_batch)
            self._static_names, self._static_dst = [k for k, _ in named], [v for _, v in named]
        else:
            frames, targets, meta = batch
            try:        # the same tensors, in the order of the static list (no sorting / renaming per batch)
                src = [frames if k == 'frames' else (targets[k[8:]] if k[0] == 't' else meta[k[5:]]) for k in self._static_names]
            except KeyError as e:
                raise RuntimeError(f'graph mode needs batches of one fixed structure: {e.args[0]!r} is missing') from None
            n_tensors = 1 + sum(isinstance(v, torch.Tensor) for v in targets.values()) + sum(isinstance(v, torch.Tensor) for v in meta.values())
            if n_tensors != len(src) or any(s.shape != d.shape or s.dtype != d.dtype for s, d in zip(src, self._static_dst)):
                raise RuntimeError('graph mode needs batches of one fixed structure: got '
                                   f'{[(k, tuple(v.shape)) for k, v in self._batch_tensors(batch)]} after '
                                   f'{[(k, tuple(v.shape)) for k, v in zip(self._static_names, self._static_dst)]}')
            if self._group_copy is None and src[0].is_cuda and all(d.is_contiguous() for d in self._static_dst):
                from pedestrians_video_2_carla_amd import ops
                self._group_copy = ops.GroupCopy(self._static_dst)           # one launch for all tensors of the batch
            if self._group_copy is not None and all(t.is_cuda and t.device == d.device for t, d in zip(src, self._static_dst)):
                self._group_copy(src)
            else:
                torch._foreach_copy_(self._static_dst, src)
            for k, v in meta.items():                     # lists of strings etc. travel by reference
                if not isinstance(v, torch.Tensor):
                    self._static_batch[2][k] = v
        self._staged_src = key
        flow.on_train_batch_start(self._static_batch, batch_idx)
        return self._static_batch

    # ------------------------------------------------------------------------------------------------------------
    def train_step(self, flow, batch, batch_idx: int = 0) -> torch.Tensor:
        """One optimisation step on ``batch``; returns the (device) loss tensor."""
        if not self.use_graph:
            loss = self._forward_backward(flow, batch, batch_idx)
            if self.exchange is not None:
                self.exchange.all_reduce_gradients()
            self._optimizer_step()
        else:
            if self._direct is None or not self._hand_over(flow, batch):
                static = self.stage_batch(flow, batch, batch_idx)      # copies only when a NEW batch object arrives
            if self._graphs is None:
                self._capture(flow, static, batch_idx)
                if not self.use_graph:              # the captured step failed its replay check: eager steps from here on
                    return self.train_step(flow, batch, batch_idx)
            g_fb, g_opt = self._graphs
            for o in self.optimizers:
                if hasattr(o, 'sync_hyper'):
                    o.sync_hyper()               # LR-scheduler changes reach the captured optimizer launch
            if self._direct is not None:            # the step IS one recorded call: two launches, no graph start-up
                d = self._direct
                if torch.cuda.current_device() != d['device_index']:   # the recorded pointers belong to the trainer's GPU
                    with torch.cuda.device(d['device_index']):
                        rc = d['call'](d['desc_ref'], d['gl'], torch.cuda.current_stream(d['device_index']).cuda_stream)
                else:
                    rc = d['call'](d['desc_ref'], d['gl'], torch.cuda.current_stream().cuda_stream)
                if rc != 0:
                    raise RuntimeError(f'p2c_train_step failed in direct replay (rc={rc})')
            else:
                g_fb.replay()
            if g_opt is not None:
                self.exchange.all_reduce_gradients()
                if g_opt == 'eager':
                    self._optimizer_step()
                else:
                    g_opt.replay()
            loss = self._static_loss
        self.global_step += 1
        flow.global_step = self.global_step
        if self.lr_schedulers:
            self.step_lr_schedulers('step')
        return loss

    # ---- direct replay: a new batch is handed over by ADDRESS ---------------------------------------------------------------
    def _handover_info(self, flow, static_batch):
        """What the recorded p2c_train_step call reads from a batch: (targets key of the 2-D target or None, whether the 3-D
        target is read) and the tensors' shapes -- worked out once, on the static batch the step was captured on."""
        frames, targets, meta = static_batch
        plan = flow._fused_train_plan(frames, targets) if hasattr(flow, '_fused_train_plan') else None
        st = meta.get('skel_type') if isinstance(meta, dict) else None
        if plan is None or not isinstance(st, torch.Tensor):
            return None
        _spec, gt2d, gt3d = plan
        key2 = next((k for k, v in targets.items() if v is gt2d), None) if gt2d is not None else None
        key3 = next((k for k, v in targets.items() if v is gt3d), None) if gt3d is not None else None
        if (gt2d is not None and key2 is None) or (gt3d is not None and key3 is None):
            return None
        d = self._direct['desc']
        # the recorded call must read the static batch's OWN tensors: a converted copy inside the recorded step would be swapped out
        # silently by a hand-over
        want = (frames.data_ptr(), gt2d.data_ptr() if gt2d is not None else d.head.gt2d,
                gt3d.data_ptr() if gt3d is not None else d.head.gt3d, st.data_ptr())
        if (d.mlp.x, d.head.gt2d, d.head.gt3d, d.head.skel_type) != want:
            return None
        return {'key2': key2, 'key3': key3, 'device': frames.device,
                'shapes': (tuple(frames.shape), tuple(gt2d.shape) if gt2d is not None else None,
                           tuple(gt3d.shape) if gt3d is not None else None, tuple(st.shape)),
                'static_ptrs': (d.mlp.x, d.head.gt2d, d.head.gt3d, d.head.skel_type)}

    def _hand_over(self, flow, batch) -> bool:
        """Direct replay only (the captured step is ONE recorded C-ABI call on a descriptor this trainer owns): a new batch is
        not copied into the static buffers -- the descriptor's four input addresses (frames, the two targets, the skeleton-type
        index) are pointed at the batch's own tensors and the target pairs are counted into the buffer the call reads. What the
        reference's loop does per batch -- move it to the device, run the batch-start hook (projection.py:52-71) -- is then
        one count launch; the batch object is kept alive until the next one arrives. Returns False when the batch does not
        have the static batch's exact layout on the device (the copying path then normalises it, as before)."""
        if batch is self._staged_src:
            return True
        h = self._direct.get('handover')
        if h is None:
            return False

        def refuse():                              # the copying path takes over: the call reads the static buffers again
            if self._handed_over is not None:
                d = self._direct['desc']
                d.mlp.x, d.head.gt2d, d.head.gt3d, d.head.skel_type = h['static_ptrs']
                self._handed_over = None
            return False
        try:
            frames, targets, meta = batch
            gt2d = targets[h['key2']] if h['key2'] is not None else None
            gt3d = targets[h['key3']] if h['key3'] is not None else None
            st = meta['skel_type']
        except (KeyError, TypeError, ValueError):
            return refuse()
        dev, f32 = h['device'], torch.float32
        for t, shape, dt in ((frames, h['shapes'][0], f32), (gt2d, h['shapes'][1], f32), (gt3d, h['shapes'][2], f32),
                             (st, h['shapes'][3], torch.int32)):
            if shape is None:
                continue
            if not (isinstance(t, torch.Tensor) and t.device == dev and t.dtype == dt and tuple(t.shape) == shape
                    and t.is_contiguous() and t.data_ptr() % 16 == 0):      # (p2c_train_step_launch wants 16-byte aligned rows)
                return refuse()
        d = self._direct['desc']
        d.mlp.x, d.head.skel_type = frames.data_ptr(), st.data_ptr()
        if gt2d is not None:
            d.head.gt2d = gt2d.data_ptr()
            flow._pair_counter(gt2d)               # into flow._pair_counts_buf, the address the recorded call reads
        if gt3d is not None:
            d.head.gt3d = gt3d.data_ptr()
        self._handed_over = batch                  # keeps the tensors alive while the launches that read them run
        self._staged_src = batch
        return True

    def _capture(self, flow, batch, batch_idx):
        """Capture on the trainer's static batch (``stage_batch`` fills it; the flow's batch-start hook already ran)."""
        distributed = self.exchange is not None and self.exchange.enabled
        # the captured launches keep the ADDRESS of the per-batch skeleton-type index: it must be the static batch's own tensor
        # (stage_batch refills it in place), not a temporary the batch-start hook derived from something else
        st = batch[2].get('skel_type') if isinstance(batch[2], dict) else None
        held = getattr(getattr(flow, 'projection', None), '_skel_type', None)
        if isinstance(st, torch.Tensor) and isinstance(held, torch.Tensor) and held.data_ptr() != st.data_ptr():
            raise RuntimeError('graph capture: ProjectionModule holds a skeleton-type tensor that is not the static batch\'s '
                               f'(meta[\'skel_type\'] is {st.dtype} on {st.device}; int32 on the frames\' device is kept as is)')
        # the warm-up iterations below are real optimisation steps: snapshot parameters + optimizer state and restore
        # them IN PLACE afterwards (the graphs hold the addresses), so that replay #1 is training step #1
        snapshot = self._snapshot(flow)
        # ONE side stream for the warm-up steps and for every capture below: autograd's AccumulateGrad nodes remember the stream
        # they were created on and outlive a step whenever anything still references its graph; a node from the warm-up that
        # runs on another stream than the capture's put the gradients of the parameters that reach the flat buffer through
        # AccumulateGrad (PoseFormer's position embeddings, patch embedding, frame-mean bias) one step late in the replayed
        # graph -- found with NaN-filled torch.empty (tests/conftest.py P2C_POISON_EMPTY), visible without it as parameters that
        # drift from the eager trainer by ~lr per step
        side = self._capture_stream = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up outside capture (allocator, lazy inits, autotuning)
            for _ in range(3):
                self._forward_backward(flow, batch, batch_idx, batch_start=False)
                if distributed:
                    self.exchange.all_reduce_gradients()
                self._optimizer_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        mode = os.environ.get('P2C_GRAPH_ALLREDUCE', 'auto')
        g_fb = self._new_graph(keep=distributed)
        if distributed and mode != '0':
            # the RCCL all-reduce is captured too: the whole step is ONE graph replay (one-rank rehearsal: 49 us against
            # 59 us for graph + eager collective + optimizer launch). Guarded: a failed capture / first replay on any rank,
            # or ranks whose parameters differ after that replay, send every rank to the eager-collective path below.
            if self._capture_with_allreduce(flow, batch, batch_idx, g_fb, snapshot, strict=(mode == '1')):
                return
            g_fb = self._new_graph(keep=True)
        if distributed:
            with torch.cuda.graph(g_fb, stream=side):
                self._static_loss = self._forward_backward(flow, batch, batch_idx, batch_start=False)
            if self._kept_graph:
                self._graph_nodes = self._count_nodes(g_fb)      # stage A alone: forward + backward
                g_fb.instantiate()
            if all(hasattr(o, '_descriptor') for o in self.optimizers):
                g_opt = 'eager'          # FlatAdamW is a single kernel: a direct launch has less latency than a 1-node graph
            else:
                g_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_opt, stream=side):
                    self._optimizer_step()
        else:
            g_opt = None
            # A step that is nothing but ONE recorded C-ABI call (the two-launch train step with the optimizer inside) is
            # replayed by making that call again: a graph launch costs ~5 us of start-up per replay, two direct kernel
            # launches ~2 us of gap (B = 256: 32.6 -> see DESIGN section 5). Checked, not assumed: the call is recorded during
            # the capture and the captured graph must hold exactly its two kernel nodes. P2C_DIRECT_REPLAY=0 keeps the graph.
            from pedestrians_video_2_carla_amd import ops
            try_direct = self._opt_in_backward and os.environ.get('P2C_DIRECT_REPLAY', '1') == '1'
            rec = [] if try_direct else None
            if try_direct:
                g_fb = self._new_graph(keep=True)
                try_direct = self._kept_graph          # an older torch without keep_graph: the plain graph replays the step
                rec = rec if try_direct else None
            ops.TRAIN_STEP_RECORDER = rec
            try:
                with torch.cuda.graph(g_fb, stream=side):
                    self._static_loss = self._forward_backward(flow, batch, batch_idx, batch_start=False)
                    self._optimizer_step()
            finally:
                ops.TRAIN_STEP_RECORDER = None
            if try_direct:
                self._direct = self._direct_replay_plan(g_fb, rec)
                g_fb.instantiate()
                if self._direct is not None and os.environ.get('P2C_HAND_OVER', '1') == '1':
                    self._direct['handover'] = self._handover_info(flow, batch)
        self._graphs = (g_fb, g_opt)
        self._restore(flow, snapshot)
        if self._direct is None and not distributed and os.environ.get('P2C_VERIFY_REPLAY', '1') == '1':
            self._verify_replay(flow, snapshot)

    def _verify_replay(self, flow, snapshot):
        """(Single-GPU captures only: the distributed captures -- all-reduce in the graph, or stage A with an eager optimizer -- are
        NOT compared with an eager step here; ``ranks_agree`` only establishes that every rank holds the same bits.)
        Capture-time check: the new graph is replayed twice -- other tensors allocated, written and freed before each replay --
        and its parameter update compared with the same step issued eagerly from the same parameters and random-number state
        (a captured step draws the same philox offsets as the eager one, and the in-kernel dropout streams are put back; the build's kernels give the same bits either way:
        measured difference exactly 0 for cfg2 / cfg3 / cfg5 with dropout and stochastic depth on). A step that reads memory the
        graph does not own, or captured a stale address, shows up here; the trainer then goes back to eager steps and says so.
        NOT caught here: on this stack (PyTorch 2.10 / ROCm 7.0) a captured backward with one of the framework's multi-block
        reductions goes wrong only later, some replays and allocations into training (tools/graph_reduce_repro.py) -- the flows
        of this build keep those out of their steps, and tests/test_graph_replay_gpu.py replays with churn. P2C_VERIFY_REPLAY=0
        skips the check."""
        g_fb, g_opt = self._graphs
        state = self.flat.flat_param.data if self.flat is not None else None
        if state is None or g_opt is not None:
            return
        dev = state.device
        rng = torch.cuda.get_rng_state(dev)
        from pedestrians_video_2_carla_amd import ops
        drop = ops.dropout_states_snapshot()                 # the masks drawn inside kernels: the same stream position each time
        before = state.clone()
        # ground truth: the same step issued eagerly on the static batch, from the same parameters and random-number state (a
        # captured step draws the same philox offsets as the eager one when it starts from the same generator state)
        torch.cuda.set_rng_state(rng, dev)
        self._forward_backward(flow, self._static_batch, 0, batch_start=False)
        self._optimizer_step()
        torch.cuda.synchronize(dev)
        want = state - before
        self._restore(flow, snapshot)
        ops.dropout_states_restore(drop)
        updates = []
        for attempt in range(2):
            # what a training loop does between two steps: other tensors come and go (and leave their values behind)
            junk = [torch.empty(1 << 20, device=dev).normal_() for _ in range(32)] + [torch.empty(256, device=dev).fill_(7.0) for _ in range(256)]
            del junk
            torch.cuda.set_rng_state(rng, dev)
            g_fb.replay()
            torch.cuda.synchronize(dev)
            updates.append(state - before)
            self._restore(flow, snapshot)
            ops.dropout_states_restore(drop)
        torch.cuda.set_rng_state(rng, dev)
        scale = float(want.abs().max())
        diff = max(float((u - want).abs().max()) for u in updates)
        self._replay_check = (diff, scale)
        if not (diff <= 0.05 * scale):                       # (NaN compares false)
            print(f'[trainer] the captured step does not replay reproducibly (its update differs from the eager step\'s by {diff:.3e} '
                  f'of {scale:.3e} after other allocations): a framework reduction in its backward? Falling back to eager steps.',
                  file=sys.stderr, flush=True)
            self._graphs = None
            self._direct = None
            self.use_graph = False

    def _new_graph(self, keep: bool = False):
        """A CUDAGraph; ``keep`` asks torch to keep the captured hipGraph_t so its nodes can be counted (instantiate() is then
        ours to call). A torch without ``keep_graph`` gives the plain graph (``self._kept_graph`` False)."""
        self._kept_graph = False
        if keep:
            try:
                g = torch.cuda.CUDAGraph(keep_graph=True)
                self._kept_graph = True
                return g
            except TypeError:
                pass
        return torch.cuda.CUDAGraph()

    @staticmethod
    def _count_nodes(graph):
        import ctypes
        from pedestrians_video_2_carla_amd import _lib
        total, kernels = ctypes.c_int32(0), ctypes.c_int32(0)
        try:
            rc = _lib.lib().p2c_graph_node_counts(graph.raw_cuda_graph(), ctypes.byref(total), ctypes.byref(kernels))
        except Exception:                       # noqa: BLE001 -- no handle on this torch
            return None
        return (total.value, kernels.value) if rc == 0 else None

    @staticmethod
    def _direct_replay_plan(graph, rec):
        """The recorded call if the captured graph is exactly its launches, else None."""
        import ctypes
        from pedestrians_video_2_carla_amd import _lib
        if rec is None or len(rec) != 1:
            return None
        total, kernels = ctypes.c_int32(0), ctypes.c_int32(0)
        try:
            rc = _lib.lib().p2c_graph_node_counts(graph.raw_cuda_graph(), ctypes.byref(total), ctypes.byref(kernels))
        except Exception:                       # noqa: BLE001 -- no handle: keep the graph
            return None
        # two launches (clip kernel + weight gradient / optimizer / losses), or three at large batches (streamed weight
        # gradient + its reduction): nothing else may sit in the graph
        if rc != 0 or total.value != kernels.value or kernels.value not in (2, 3):
            return None
        plan = rec[0]
        plan['device_index'] = plan['device'].index if plan['device'].index is not None else torch.cuda.current_device()
        plan['call'] = _lib.lib().p2c_train_step
        plan['desc_ref'] = ctypes.byref(plan['desc'])
        return plan

    def _capture_with_allreduce(self, flow, batch, batch_idx, graph, snapshot, strict: bool) -> bool:
        """Capture forward + backward + all-reduce + optimizer as one graph. The ranks VOTE before anything that carries a
        collective is replayed: a rank whose capture raised never meets a healthy rank's in-graph all-reduce with its own
        eager one (collectives of different sizes pair up and the job hangs). Only when every rank captured does the
        verification replay run, followed by a second agreement on bit-identical parameters."""
        ok, err = True, None
        try:
            with torch.cuda.graph(graph, stream=self._capture_stream):
                self._static_loss = self._forward_backward(flow, batch, batch_idx, batch_start=False)
                self.exchange.all_reduce_gradients()
                self._optimizer_step()
        except Exception as e:                  # noqa: BLE001 -- any failure means "use the eager collective"
            ok, err = False, e
        torch.cuda.synchronize()                # a capture that aborted mid-way leaves nothing in flight behind it
        device = self.flat.flat_param.device if self.flat is not None else next(flow.parameters()).device
        agreed = ranks_agree(ok, device)        # eager collectives on the (non-capturing) current stream
        if agreed:
            try:
                self._graph_nodes = self._count_nodes(graph)     # (all nodes, kernel nodes) of the captured step, or None
                if self._kept_graph:
                    graph.instantiate()
                graph.replay()                  # one real step: every rank must come out with the same parameters
                torch.cuda.synchronize()
            except Exception as e:              # noqa: BLE001
                ok, err = False, e
            params = self.flat.flat_param.data if self.flat is not None else torch.cat([p.data.reshape(-1) for p in flow.parameters()])
            agreed = ranks_agree(ok, device, params.double().sum())
        self._restore(flow, snapshot)
        if agreed:
            self._graphs = (graph, None)
            return True
        if strict:
            raise RuntimeError(f'P2C_GRAPH_ALLREDUCE=1: capturing the all-reduce failed ({err!r})')
        if dist.get_rank() == 0:
            print(f'[trainer] captured all-reduce unavailable ({err!r}); using the eager collective', file=sys.stderr, flush=True)
        return False

    def _state_tensors(self, flow):
        tensors = [p.data for p in flow.parameters()] if self.flat is None else [self.flat.flat_param.data]
        tensors += [b for b in flow.buffers()]
        for o in self.optimizers:
            for st in o.state.values():
                tensors += [v for v in st.values() if isinstance(v, torch.Tensor)]
        return tensors

    def _snapshot(self, flow):
        # optimizer state is created lazily by the first step: run one throw-away step so every state tensor exists
        # and the snapshot / restore pair can work in place
        params = [t.clone() for t in self._state_tensors(flow)]
        from pedestrians_video_2_carla_amd import ops
        self._drop_pre = ops.dropout_states_snapshot()       # the streams of the in-kernel dropout masks belong to the state too
        return params

    def _restore(self, flow, snapshot):
        tensors = self._state_tensors(flow)
        with torch.no_grad():
            for t, s in zip(tensors, snapshot):          # parameters / buffers / pre-existing optimizer state
                t.copy_(s)
            for t in tensors[len(snapshot):]:            # optimizer state created during warm-up: back to step 0
                t.zero_()
            if self.flat is not None:
                self.flat.zero_grad()                    # whatever the warm-up steps left in the gradient buffer
            from pedestrians_video_2_carla_amd import ops
            ops.dropout_states_restore(getattr(self, '_drop_pre', []), rewind_new=True)   # replay #1 draws what eager step #1 would
        for m in getattr(self, '_packed', []):           # the parameters were rewritten behind the optimizer's back
            m.repack()
        torch.cuda.synchronize()

    def fit(self, flow, datamodule, batches: Optional[Iterable] = None):
        self.setup(flow, datamodule)
        device = self.device or flow.device
        if batches is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
            batches = datamodule.train_batches(device, self.max_steps, rank=rank)
        losses = []
        for i, batch in enumerate(batches):
            if i >= self.max_steps:
                break
            losses.append(self.train_step(flow, batch, i))
            if self.log_every_n_steps and (i + 1) % self.log_every_n_steps == 0:
                flow.check_finite('train')
            if self.steps_per_epoch and (i + 1) % self.steps_per_epoch == 0:
                self.current_epoch += 1
                self.step_lr_schedulers('epoch', {k: float(v) for k, v in getattr(flow, 'logged', {}).items()
                                                  if isinstance(v, (float, int)) or (isinstance(v, torch.Tensor) and v.ndim == 0)})
        return losses
