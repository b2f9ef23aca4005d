"""Normalizer: (x - shift) / scale per frame on device (reference normalizer.py:7-49)."""
from typing import Any

import torch

from .extractor import Extractor


class Normalizer(object):
    def __init__(self, extractor: Extractor, near_zero: float = 1e-5) -> None:
        self.extractor = extractor
        self.__near_zero = near_zero
        self.__last_scale = None
        self.__last_shift = None

    def __repr__(self) -> str:
        return f'{self.__class__.__name__}(extractor={self.extractor.__class__.__name__})'

    @property
    def kind(self) -> str:
        return self.extractor.kind

    def __call__(self, sample: torch.Tensor, dim=2, *args: Any, **kwargs: Any) -> torch.Tensor:
        from pedestrians_video_2_carla_amd import ops
        hips, neck = self.extractor.points()
        out, shift, scale = ops.normalize(sample, self.extractor.kind, dim, hips, neck, self.__near_zero)
        self.__last_scale, self.__last_shift = scale, shift
        return out

    @property
    def scale(self) -> torch.Tensor:
        return self.__last_scale.clone()

    @property
    def shift(self) -> torch.Tensor:
        return self.__last_shift.clone()
