"""Shift / scale extractors of the normaliser (reference transforms/pose/normalization/*.py).

In the reference these are chains of small torch ops (index, mean, norm, clone + min/max, three ``torch.any`` host
syncs in the fallback). Here an extractor is a *description* (which joints, which rule) that the HIP kernels interpret:
``kind`` selects P2C_TRANSFORM_*, ``points()`` gives the hips / neck joint indices. ``get_shift_scale`` is kept for API
compatibility and runs the stand-alone normaliser kernel.
"""
from typing import Iterable, Tuple, Type, Union

import torch

from pedestrians_video_2_carla_amd.data.base.skeleton import Skeleton


class Extractor(object):
    kind = None   # 'hips_neck' | 'bbox' | 'hips_neck_bbox'

    def __init__(self, input_nodes: Type[Skeleton], near_zero: float = 1e-5) -> None:
        self.input_nodes = input_nodes
        self.near_zero = near_zero

    @staticmethod
    def _point_to_tuple(point: Union[Skeleton, Iterable[Skeleton]]) -> Tuple[int, ...]:
        return (point.value,) if isinstance(point, Skeleton) else tuple(p.value for p in point)

    def points(self) -> Tuple[Tuple[int, ...], Tuple[int, ...]]:
        """(hips joints, neck joints) averaged into the shift / scale points (hips_neck_extractor.py:6-13)."""
        if self.kind == 'bbox' or self.input_nodes is None:
            return (0,), (0,)
        return (self._point_to_tuple(self.input_nodes.get_hips_point()),
                self._point_to_tuple(self.input_nodes.get_neck_point()))

    def get_shift_scale(self, sample: torch.Tensor):
        """shift (..., dim), scale (...) of ``sample`` (..., joints, dim) -- extractor.py:23-36."""
        from pedestrians_video_2_carla_amd import ops
        hips, neck = self.points()
        _, shift, scale = ops.normalize(sample, self.kind, sample.shape[-1], hips, neck, self.near_zero)
        return shift, scale
