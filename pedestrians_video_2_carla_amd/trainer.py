"""Minimal trainer: the Lightning loop the reference delegates to, reduced to what the train step needs
(reference modeling.py:275-282 ``pl.Trainer.from_argparse_args(...).fit(model, datamodule)``).

Hook order per step is Lightning's: ``on_train_batch_start`` -> ``training_step`` -> backward -> optimizer step.
Two execution modes:
  * eager: every launch issued from Python;
  * ``use_graph=True``: the step (model forward, HIP pose head forward/backward, optimizer) is captured once into a
    HIP graph on static batch buffers and replayed -- the launch-bound regime at B=256 (about 50 kernels of a few
    microseconds each) is exactly what hipGraphs are for. With more than one rank the gradient all-reduce stays outside
    the graphs (capture A: zero-grad + forward + backward; eager RCCL all-reduce; capture B: optimizer).
"""
import contextlib
import os
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


class Trainer:
    def __init__(self, max_steps: int = 100, use_graph: bool = False, device: Optional[torch.device] = None,
                 flatten: bool = True, log_every_n_steps: int = 0):
        self.max_steps = max_steps
        self.use_graph = use_graph
        self.device = device
        self.flatten = flatten
        self.log_every_n_steps = log_every_n_steps
        self.datamodule = None
        self.loggers: List = []
        self.global_step = 0
        self.flat: Optional[FlatParameters] = None
        self.exchange: Optional[GradientExchange] = None
        self.optimizers: List[torch.optim.Optimizer] = []
        self._graphs = None
        self._static_loss = None
        self._unit = None
        self._packed = []
        self._opt_in_backward = False

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
            extra = {}
            if self.flat.flat_param.is_cuda:
                extra['fused'] = True              # one kernel for the whole (flat) parameter
                extra['capturable'] = bool(self.use_graph)
                # plugins with a fused backward write their parameter gradients straight into the flat buffer
                for m in flow.modules():
                    if hasattr(m, 'grad_sink'):
                        m.grad_sink = True
                from pedestrians_video_2_carla_amd import ops as _ops
                _ops.GRAD_SINKS = True           # K12 adds weight gradients straight into the flat gradient buffer's views
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
            if hasattr(opt, 'grad_scale') and self.exchange.enabled:    # FlatAdamW folds the DP averaging into its pass
                opt.grad_scale = 1.0 / self.exchange.world
                self.exchange.average_here = False
        else:
            self.optimizers = [c['optimizer'] for c in configs]
        return self

    def _zero_grad(self):
        if self.flat is not None:
            if getattr(self.optimizers[0], 'zero_grad_in_step', False):
                return                  # the flat gradient starts zeroed and FlatAdamW leaves it zeroed after each step
            self.flat.zero_grad()
        else:
            for o in self.optimizers:
                o.zero_grad(set_to_none=True)

    def _forward_backward(self, flow, batch, batch_idx):
        self._zero_grad()
        # forward and backward happen back to back in here and nothing reads a loss VALUE in between (the reference's
        # per-loss isnan check is off unless strict_nan_check): the pose head may leave the last stage of its loss
        # reduction to the backward kernel, or run only in the backward at all (P2C_DEFER_FINALIZE=0|1|2, see
        # ops.deferred_loss_finalize)
        mode = int(os.environ.get('P2C_DEFER_FINALIZE', '2'))      # 2: the pose head runs once per step, in the backward
        defer = batch[0].is_cuda and not getattr(flow, 'strict_nan_check', False) and mode != 0
        from pedestrians_video_2_carla_amd import ops
        with (ops.deferred_loss_finalize(mode) if defer else contextlib.nullcontext()):
            flow.on_train_batch_start(batch, batch_idx)
            out = flow.training_step(batch, batch_idx)
            loss = out['loss']
            if self._unit is None or self._unit.shape != loss.shape or self._unit.device != loss.device:
                self._unit = torch.ones_like(loss)   # root gradient, made once (backward() would fill one per step)
            loss.backward(gradient=self._unit)
        return loss.detach()

    def _optimizer_step(self):
        if self._opt_in_backward:       # already applied by the backward launch of the module that owns the parameters
            return
        for o in self.optimizers:
            o.step()

    # ------------------------------------------------------------------------------------------------------------
    def train_step(self, flow, batch, batch_idx: int = 0) -> torch.Tensor:
        """One optimisation step on ``batch``; returns the (device) loss tensor."""
        if not self.use_graph:
            loss = self._forward_backward(flow, batch, batch_idx)
            if self.exchange is not None:
                self.exchange.all_reduce_gradients()
            self._optimizer_step()
        else:
            if self._graphs is None:
                self._capture(flow, batch, batch_idx)
            g_fb, g_opt = self._graphs
            for o in self.optimizers:
                if hasattr(o, 'sync_hyper'):
                    o.sync_hyper()               # LR-scheduler changes reach the captured optimizer launch
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
        return loss

    def _capture(self, flow, batch, batch_idx):
        """Capture on the given batch: its tensors become the static input buffers (copy new data into them)."""
        distributed = self.exchange is not None and self.exchange.enabled
        # the warm-up iterations below are real optimisation steps: snapshot parameters + optimizer state and restore
        # them IN PLACE afterwards (the graphs hold the addresses), so that replay #1 is training step #1
        snapshot = self._snapshot(flow)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up outside capture (allocator, lazy inits, autotuning)
            for _ in range(3):
                self._forward_backward(flow, batch, batch_idx)
                if distributed:
                    self.exchange.all_reduce_gradients()
                self._optimizer_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g_fb = torch.cuda.CUDAGraph()
        mode = os.environ.get('P2C_GRAPH_ALLREDUCE', 'auto')
        if distributed and mode != '0':
            # the RCCL all-reduce is captured too: the whole step is ONE graph replay (one-rank rehearsal: 49 us against
            # 59 us for graph + eager collective + optimizer launch). Guarded: a failed capture / first replay on any rank,
            # or ranks whose parameters differ after that replay, send every rank to the eager-collective path below.
            if self._capture_with_allreduce(flow, batch, batch_idx, g_fb, snapshot, strict=(mode == '1')):
                return
            g_fb = torch.cuda.CUDAGraph()
        if distributed:
            with torch.cuda.graph(g_fb):
                self._static_loss = self._forward_backward(flow, batch, batch_idx)
            if all(hasattr(o, '_descriptor') for o in self.optimizers):
                g_opt = 'eager'          # FlatAdamW is a single kernel: a direct launch has less latency than a 1-node graph
            else:
                g_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_opt):
                    self._optimizer_step()
        else:
            g_opt = None
            with torch.cuda.graph(g_fb):
                self._static_loss = self._forward_backward(flow, batch, batch_idx)
                self._optimizer_step()
        self._graphs = (g_fb, g_opt)
        self._restore(flow, snapshot)

    def _capture_with_allreduce(self, flow, batch, batch_idx, graph, snapshot, strict: bool) -> bool:
        import torch.distributed as dist
        ok, err = True, None
        try:
            with torch.cuda.graph(graph):
                self._static_loss = self._forward_backward(flow, batch, batch_idx)
                self.exchange.all_reduce_gradients()
                self._optimizer_step()
            graph.replay()                      # one real step: every rank must come out with the same parameters
            torch.cuda.synchronize()
        except Exception as e:                  # noqa: BLE001 -- any failure means "use the eager collective"
            ok, err = False, e
        device = self.flat.flat_param.device if self.flat is not None else next(flow.parameters()).device
        params = self.flat.flat_param.data if self.flat is not None else torch.cat([p.data.reshape(-1) for p in flow.parameters()])
        chk = params.double().sum() if ok else torch.zeros((), dtype=torch.float64, device=device)
        votes = torch.stack((torch.tensor(1.0 if ok else 0.0, dtype=torch.float64, device=device), -chk, chk))
        lo = votes.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)           # eager collectives: agreement on the outcome
        hi = votes.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        agreed = bool(lo[0] > 0.5) and bool(torch.isfinite(chk)) and float(hi[2]) == float(lo[2])
        self._restore(flow, snapshot)
        if agreed:
            self._graphs = (graph, None)
            return True
        if strict:
            raise RuntimeError(f'P2C_GRAPH_ALLREDUCE=1: capturing the all-reduce failed ({err!r})')
        if dist.get_rank() == 0:
            print(f'[trainer] captured all-reduce unavailable ({err!r}); using the eager collective', flush=True)
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
        return params

    def _restore(self, flow, snapshot):
        tensors = self._state_tensors(flow)
        with torch.no_grad():
            for t, s in zip(tensors, snapshot):          # parameters / buffers / pre-existing optimizer state
                t.copy_(s)
            for t in tensors[len(snapshot):]:            # optimizer state created during warm-up: back to step 0
                t.zero_()
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
        return losses
