from .extractor import Extractor


class BBoxExtractor(Extractor):
    """shift = centre of the bounding box of the detected joints, scale = half its height
    (reference bbox_extractor.py:6-18, utils/tensors.py:12-26)."""
    kind = 'bbox'
