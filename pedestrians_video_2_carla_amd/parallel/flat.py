"""Flat parameter / gradient buffers and the one-collective-per-step gradient exchange.

The reference trains under Lightning DDP (README.md:74-75 ``--gpus=0,1 --accelerator=ddp``): bucketed all-reduce hooks
on ~12-100 small tensors, parameter broadcast at wrap time. The payloads here are tiny
(LinearAE: 17 530 fp32 = 70 KB; Seq2SeqEmbeddings: 2.2 MB) -- latency-bound on xGMI, so:

  * all trainable parameters are re-seated as views into ONE contiguous fp32 buffer, all ``.grad`` as views into
    another; autograd accumulates straight into the flat gradient buffer;
  * one ``all_reduce(SUM)`` on that buffer per step (RCCL when the process group is ``nccl``; ``gloo`` on CPU),
    followed by a 1/world scale fused into the same buffer -- DDP's mean-of-per-rank-means semantics;
  * the optimizer runs on the flat parameter as a single tensor (AdamW is element-wise, so the update is identical),
    i.e. one fused kernel instead of a multi-tensor loop.
"""
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class FlatParameters:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params), 'parameters must share device and dtype'
        sizes = [p.numel() for p in self.params]
        self.numel = sum(sizes)
        flat = torch.empty(self.numel, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(self.numel, device=dev, dtype=dt)
        off = 0
        for p, n in zip(self.params, sizes):
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view_as(p.data)
            p.grad = self.flat_grad[off:off + n].view_as(p.data)
            off += n
        self.flat_param = torch.nn.Parameter(flat, requires_grad=True)
        self.flat_param.grad = self.flat_grad

    def zero_grad(self):
        self.flat_grad.zero_()          # one memset; keeps the .grad views alive (set_to_none would detach them)

    def nbytes(self) -> int:
        return self.numel * self.flat_grad.element_size()

    def rebuild_optimizer(self, optimizer: torch.optim.Optimizer, **overrides) -> torch.optim.Optimizer:
        """Same optimizer class / hyper-parameters, but over the single flat parameter."""
        if len(optimizer.param_groups) != 1:
            raise ValueError('flat optimizer needs a single param group (the reference configures exactly one)')
        g0 = optimizer.param_groups[0]
        if (type(optimizer) in (torch.optim.AdamW, torch.optim.Adam) and self.flat_param.is_cuda
                and not g0.get('amsgrad', False) and not g0.get('maximize', False) and overrides.pop('native', True)):
            from pedestrians_video_2_carla_amd.parallel.optim import FlatAdamW      # one HIP launch, graph-safe
            overrides.pop('fused', None), overrides.pop('capturable', None)
            decoupled = type(optimizer) is torch.optim.AdamW or bool(g0.get('decoupled_weight_decay', False))
            return FlatAdamW([self.flat_param], lr=g0['lr'], betas=g0['betas'], eps=g0['eps'],
                             weight_decay=g0['weight_decay'], decoupled=decoupled)
        overrides.pop('native', None)
        import inspect
        accepted = set(inspect.signature(type(optimizer).__init__).parameters)
        group = {k: v for k, v in optimizer.param_groups[0].items() if k != 'params' and k in accepted}
        group.update(overrides)
        return type(optimizer)([self.flat_param], **group)


class GradientExchange:
    """Data-parallel gradient averaging: ONE all-reduce of the flat gradient buffer per step."""

    def __init__(self, flat: FlatParameters, process_group: Optional[dist.ProcessGroup] = None):
        self.flat = flat
        self.group = process_group
        self.average_here = True        # False: the optimizer kernel applies the 1/world factor (FlatAdamW.grad_scale)
        initialised = dist.is_available() and dist.is_initialized()
        # P2C_FORCE_EXCHANGE=1 keeps the collective (and the two-graph step) with a single rank: lets the one-GPU box
        # exercise exactly the code path the multi-GPU runs take
        forced = initialised and os.environ.get('P2C_FORCE_EXCHANGE', '0') == '1'
        self.enabled = initialised and (dist.get_world_size(process_group) > 1 or forced)
        self.world = dist.get_world_size(process_group) if self.enabled else 1

    def broadcast_parameters(self, src: int = 0):
        if self.enabled:
            dist.broadcast(self.flat.flat_param.data, src=src, group=self.group)

    def broadcast_buffers(self, module: torch.nn.Module, src: int = 0):
        """Module buffers (BatchNorm running statistics of LinearAEResidual ...) from rank ``src``: what Lightning's DDP
        wrapper does at wrap time (``broadcast_buffers=True``). Per-step re-synchronisation is NOT done: with identical
        initial buffers the ranks' statistics differ only by their own shards' batches, as under SyncBatchNorm-less DDP
        between two forward passes."""
        if self.enabled:
            for b in module.buffers():
                dist.broadcast(b, src=src, group=self.group)

    def all_reduce_gradients(self):
        if self.enabled:
            dist.all_reduce(self.flat.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            if self.average_here:
                self.flat.flat_grad.mul_(1.0 / self.world)


def all_reduce_loss_sums(sum_and_count: torch.Tensor, group=None) -> torch.Tensor:
    """Exact global masked mean: all-reduce (sum, count) pairs instead of per-rank means."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sum_and_count, op=dist.ReduceOp.SUM, group=group)
    return sum_and_count
