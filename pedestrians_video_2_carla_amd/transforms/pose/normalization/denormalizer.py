"""DeNormalizer: x * scale + shift with the reference's broadcasting rules (denormalizer.py:7-33).

Two fused multiply-adds per element; kept as differentiable tensor expressions because on the training hot path the
only caller (absolute_loc outputs, modules/layers/projection.py:125-136) is folded into the HIP pose head.
"""
from typing import Any, Callable

import torch

from .extractor import Extractor


class DeNormalizer(object):
    def __call__(self, sample: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, dim=2, *args: Any,
                 **kwargs: Any) -> torch.Tensor:
        d = scale[(slice(None),) * scale.ndim + (None,) * (sample.ndim - scale.ndim)]
        h = shift[(slice(None),) * scale.ndim + (None,) * (sample.ndim - shift.ndim) + (slice(None),)]
        head = sample[..., 0:dim] * d + h
        if dim == 2 and sample.shape[-1] > 2:
            return torch.cat((head, sample[..., 2:3], sample[..., 3:]), dim=-1)
        return head if sample.shape[-1] == dim else torch.cat((head, torch.empty_like(sample[..., dim:])), dim=-1)

    @staticmethod
    def from_reference(extractor: Extractor, reference: torch.Tensor) -> Callable:
        shift, scale = extractor.get_shift_scale(reference)
        instance = DeNormalizer()
        return lambda sample, dim=2: instance(sample, scale, shift, dim)
