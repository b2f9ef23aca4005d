"""LinearAE movements model (reference modules/movements/linear_ae/linear_ae.py:5-59).

Per-frame MLP 2P -> P -> P/2 -> P/4 -> O/4 -> O/2 -> O with ReLU between (P = input joints * 2, O = output joints *
output_features). Attribute names are kept (``__encoder`` / ``__decoder`` inside class ``LinearAE``) so state_dict keys
(``_LinearAE__encoder.0.weight`` ...) match reference checkpoints.

On the GPU the six Linear(+ReLU) layers run as ONE fp32-MFMA kernel forward and one backward (csrc/p2c_mlp.hip,
``ops.fused_mlp``) instead of ~50 ATen launches; ``fused_mlp=False`` (or host tensors, which the plugin still accepts for
checkpoint / golden-vector work) uses the plain ``nn.Sequential``. Both paths compute the same fp32 function.
"""
import torch
from torch import nn

from pedestrians_video_2_carla_amd.modules.movements.movements import MovementsModel, MovementsModelOutputTypeMixin


def _mlp(sizes, last_activation):
    layers = []
    for i, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
        layers.append(nn.Linear(a, b))
        if last_activation or i < len(sizes) - 2:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class LinearAE(MovementsModelOutputTypeMixin, MovementsModel):
    def __init__(self, fused_mlp: bool = True, mlp_precision: str = None, **kwargs):
        super().__init__(**kwargs)
        self.fused_mlp = fused_mlp
        # operand precision of the fused MLP's matrix products: 'fp32' (exact, default), 'bf16', 'bf16x3' (ops.fused_mlp);
        # P2C_MLP_PRECISION sets the default. The reduced arms are opt-in: every parity claim is made with fp32.
        import os
        self.mlp_precision = mlp_precision or os.environ.get('P2C_MLP_PRECISION', 'fp32')
        self.grad_sink = False          # set by the trainer: write parameter gradients straight into .grad
        self._image = None              # persistent packed-weight image of the fused MLP (device buffer, not a parameter)
        self._image_managed = False     # True: an optimizer keeps the image current -> no pack launch in forward()
        self.fused_optimizer = None     # set by the single-GPU trainer: the optimizer step rides on this module's backward
        self.__n_out = len(self.output_nodes)
        self.__in = len(self.input_nodes) * 2                  # (x, y) per joint
        out = self.__n_out * self.output_features
        self.__encoder = _mlp([self.__in, self.__in // 2, self.__in // 4, self.__in // 8], last_activation=True)
        self.__decoder = _mlp([self.__in // 8, out // 4, out // 2, out], last_activation=False)

    @staticmethod
    def add_model_specific_args(parent_parser):
        parent_parser = MovementsModel.add_model_specific_args(parent_parser)
        MovementsModelOutputTypeMixin.add_cli_args(parent_parser.add_argument_group('LinearAE Model'))
        return parent_parser

    def _linears(self):
        return [m for m in list(self.__encoder) + list(self.__decoder) if isinstance(m, nn.Linear)]

    # ---- packed weight image kept current by the optimizer (Trainer + FlatAdamW) -----------------------------------------
    def manage_packed_image(self, flat_param: torch.Tensor, optimizer) -> bool:
        """Let ``optimizer`` (FlatAdamW over ``flat_param``, of which the Linear weights are views) write every updated
        weight also into the fused MLP's packed image: forward() then launches no pack kernel. Anything else that changes
        the weights (``load_state_dict`` is hooked; manual ``.data`` edits are not) must be followed by ``repack()``."""
        from pedestrians_video_2_carla_amd import ops
        layers = self._linears()
        dims = [self.__in] + [m.out_features for m in layers]
        if not (self.fused_mlp and flat_param.is_cuda and ops.mlp_supported(dims) and hasattr(optimizer, 'set_scatter')):
            return False
        n_image, index = ops.mlp_image_layout(dims)
        scatter = torch.full((flat_param.numel(),), -1, dtype=torch.int32)
        pos, base, esz = 0, flat_param.data_ptr(), flat_param.element_size()
        for m in layers:
            for p in (m.weight, m.bias):
                off = (p.data_ptr() - base) // esz
                if off < 0 or off + p.numel() > flat_param.numel() or not p.is_contiguous():
                    return False                      # not a view of the flat buffer: keep packing in forward()
                scatter[off:off + p.numel()] = index[pos:pos + p.numel()]
                pos += p.numel()
        self._image = torch.empty(n_image, dtype=torch.float32, device=flat_param.device)
        optimizer.set_scatter(scatter.to(flat_param.device), self._image)
        self._image_managed = True
        self.repack()
        if not getattr(self, '_repack_hook', None):
            self._repack_hook = self.register_load_state_dict_post_hook(lambda module, keys: module.repack())
        return True

    def accept_fused_optimizer(self, optimizer, flat_param: torch.Tensor) -> bool:
        """Single-GPU training: let the backward of this module apply ``optimizer``'s step inside its gradient reduction
        (one launch less per step). Only valid when this module's Linear layers are ALL the optimizer optimises."""
        layers = self._linears()
        n = sum(m.weight.numel() + m.bias.numel() for m in layers)
        if not (self._image_managed and hasattr(optimizer, 'descriptor_for_fusion') and n == flat_param.numel()):
            return False
        self.fused_optimizer = optimizer
        return True

    def repack(self):
        if self._image_managed and self._image is not None:
            from pedestrians_video_2_carla_amd import ops
            layers = self._linears()
            ops.mlp_pack([m.weight for m in layers], [m.bias for m in layers], self._image)

    def fused_args(self, device):
        """What ``ops.fused_mlp`` / ``ops.fused_train_step`` need from this module in its current state (weights, gradient
        sinks, packed image, optimizer riding on the backward), or None when the ATen path has to run."""
        from pedestrians_video_2_carla_amd import ops
        layers = self._linears()
        dims = [self.__in] + [m.out_features for m in layers]
        if not (self.fused_mlp and ops.mlp_supported(dims)):
            return None
        sinks = None
        if self.grad_sink and torch.is_grad_enabled() and all(m.weight.grad is not None for m in layers):
            sinks = [g for m in layers for g in (m.weight.grad, m.bias.grad)]
        managed = self._image_managed and self._image is not None and self._image.device == device
        fused_opt = self.fused_optimizer if (sinks is not None and self.training) else None
        return dict(dims=dims, weights=[m.weight for m in layers], biases=[m.bias for m in layers], sinks=sinks,
                    image=self._image if managed else None, image_is_current=managed, fused_optimizer=fused_opt)

    def forward(self, x, *args, **kwargs):
        lead = x.shape[0:2]
        flat = x.reshape((-1, self.__in))
        if self.fused_mlp and flat.is_cuda and flat.dtype == torch.float32:
            from pedestrians_video_2_carla_amd import ops
            fa = self.fused_args(flat.device)
            if fa is not None:
                h = ops.fused_mlp(flat, fa['weights'], fa['biases'], fa['sinks'], image=fa['image'],
                                  image_is_current=fa['image_is_current'], fused_optimizer=fa['fused_optimizer'],
                                  precision=self.mlp_precision)
                return self._format_output(h.view(*lead, self.__n_out, self.output_features))
        h = self.__decoder(self.__encoder(flat))
        return self._format_output(h.view(*lead, self.__n_out, self.output_features))
