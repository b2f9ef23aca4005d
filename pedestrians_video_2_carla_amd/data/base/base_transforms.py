from enum import Enum


class BaseTransforms(Enum):
    """Data-module transforms (reference data/base/base_transforms.py:4-10)."""
    none = 0
    hips_neck = 1
    bbox = 2
    hips_neck_bbox = 3
    user_defined = 100
