"""De-normalise to the scale of the clip's CARLA reference skeleton (reference reference_skeletons_denormalizer.py:32-91)."""
from typing import Any, Dict, List

import torch
from torch import Tensor

from pedestrians_video_2_carla_amd.data.carla import reference as ref
from pedestrians_video_2_carla_amd.data.carla.reference import AGE_MAPPINGS, GENDER_MAPPINGS  # noqa: F401
from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
from .denormalizer import DeNormalizer
from .extractor import Extractor
from .hips_neck_extractor import HipsNeckExtractor
from .normalizer import Normalizer


class ReferenceSkeletonsDeNormalizer(DeNormalizer):
    def __init__(self, extractor: Extractor = None) -> None:
        self._extractor = extractor if extractor is not None else HipsNeckExtractor(CARLA_SKELETON)
        self._normalizer = Normalizer(self._extractor)

    def _types(self, meta: Dict[str, List[Any]], n: int, device) -> Tensor:
        return ref.skeleton_types_from_meta(meta, batch_size=n, device=device).long()

    def from_projection(self, frames: Tensor, meta: Dict[str, List[Any]], autonormalize: bool = False) -> Tensor:
        if autonormalize:
            frames = self._normalizer(frames, dim=2)
        proj = ref.get_projections(frames.device)[self._types(meta, len(frames), frames.device)]
        return self.from_reference(self._extractor, proj[..., :2].contiguous())(frames, dim=2)

    def from_abs(self, frames: Tensor, meta: Dict[str, List[Any]], autonormalize: bool = False) -> Tensor:
        if autonormalize:
            frames = self._normalizer(frames, dim=3)
        abs_loc = ref.get_absolute_tensors(frames.device)[0][self._types(meta, len(frames), frames.device)]
        return self.from_reference(self._extractor, abs_loc.contiguous())(frames, dim=3)
