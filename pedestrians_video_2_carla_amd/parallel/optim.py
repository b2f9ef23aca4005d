"""``FlatAdamW``: torch.optim.AdamW / Adam semantics, one HIP launch over the flat parameter buffer.

The reference configures ``torch.optim.AdamW(self.parameters(), lr, weight_decay)`` (modules/flow/base_model.py:156-158).
With every parameter a view of one flat buffer (parallel/flat.py) the step is a single element-wise pass
(``p2c_adamw_step``, csrc/p2c_optim.hip). Differences from handing the flat tensor to torch's fused AdamW:
  * step counter and hyper-parameters live in device memory -> the launch can be captured in a HIP graph and still follow
    an LR scheduler (``param_groups[0]['lr']`` is re-uploaded when it changes);
  * the 1/world_size gradient averaging of data-parallel training (``grad_scale``) and next step's ``zero_grad`` are
    folded into the same pass.
``state_dict()`` has torch's layout (``step``, ``exp_avg``, ``exp_avg_sq``) for ONE flat parameter: it is interchangeable with a
torch.optim.AdamW built over the same flat tensor, NOT with the reference's per-parameter optimizer state (there the moments
are one pair per module parameter; slicing the flat moments by ``FlatParameters.params`` offsets converts between the two).
"""
import ctypes

import torch

from pedestrians_video_2_carla_amd import _lib


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, decoupled=True,
                 zero_grad_in_step=True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        ps = [p for g in self.param_groups for p in g['params']]
        if len(self.param_groups) != 1 or len(ps) != 1:
            raise ValueError('FlatAdamW optimises exactly one flat parameter tensor')
        (p,) = ps
        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
            raise _lib.P2CError('FlatAdamW needs a contiguous fp32 parameter on the GPU (there is no CPU fallback)')
        self.decoupled = bool(decoupled)
        self.zero_grad_in_step = bool(zero_grad_in_step)
        self.grad_scale = 1.0
        self.fused_steps_applied = 0              # bumped by every backward that applied this optimizer's step itself
        self._uploaded = None
        self._scatter = None                      # (int32 index per parameter, destination buffer) or None
        self._hyper = torch.zeros(6, dtype=torch.float32, device=p.device)
        self._ticket = torch.zeros(1, dtype=torch.int32, device=p.device)
        self.state[p] = {'step': torch.zeros((), dtype=torch.float32, device=p.device),
                         'exp_avg': torch.zeros_like(p, memory_format=torch.preserve_format),
                         'exp_avg_sq': torch.zeros_like(p, memory_format=torch.preserve_format)}

    def set_scatter(self, index: torch.Tensor, dst: torch.Tensor):
        """After every step, parameter i is also written to ``dst[index[i]]`` (index < 0: not copied)."""
        (p,) = self.param_groups[0]['params']
        if index.numel() != p.numel() or index.dtype != torch.int32 or not index.is_cuda or not dst.is_cuda:
            raise ValueError('scatter index: one int32 per parameter, on the device')
        self._scatter = (index.contiguous(), dst)

    def _descriptor(self, p):
        """The launch descriptor only holds addresses of long-lived buffers: built once, rebuilt if one of them moves."""
        st = self.state[p]
        key = (p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), st['step'].data_ptr(),
               None if self._scatter is None else (self._scatter[0].data_ptr(), self._scatter[1].data_ptr()),
               self.decoupled, self.zero_grad_in_step)
        if getattr(self, '_desc_key', None) != key:
            d = _lib.AdamWDesc()
            d.n = p.numel()
            d.param, d.grad = p.data_ptr(), p.grad.data_ptr()
            d.exp_avg, d.exp_avg_sq, d.step = st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), st['step'].data_ptr()
            d.ticket, d.hyper = self._ticket.data_ptr(), self._hyper.data_ptr()
            d.adamw, d.zero_grad = int(self.decoupled), int(self.zero_grad_in_step)
            if self._scatter is not None:
                d.scatter_idx, d.scatter_dst = self._scatter[0].data_ptr(), self._scatter[1].data_ptr()
            self._desc, self._desc_key = d, key
        return self._desc

    def descriptor_for_fusion(self):
        """Launch descriptor for a kernel that applies this optimizer's step itself (p2c_mlp_desc.fused_adamw)."""
        (p,) = self.param_groups[0]['params']
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        elif self._uploaded is None:
            raise RuntimeError('FlatAdamW: call sync_hyper() (or one eager step) before capturing a graph')
        return self._descriptor(p)

    def sync_hyper(self):
        """Upload lr / betas / eps / weight_decay / grad_scale if they changed on the host (call outside graph replay)."""
        g = self.param_groups[0]
        values = (float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay']),
                  float(self.grad_scale))
        if values != self._uploaded:
            self._hyper.copy_(torch.tensor(values, dtype=torch.float32))
            self._uploaded = values

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        (p,) = self.param_groups[0]['params']
        if p.grad is None:
            return loss
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        elif self._uploaded is None:
            raise RuntimeError('FlatAdamW: call sync_hyper() (or one eager step) before capturing a graph')
        d = self._descriptor(p)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().p2c_adamw_step(ctypes.byref(d), torch.cuda.current_stream(p.device).cuda_stream),
                       'p2c_adamw_step')
        return loss
