from .extractor import Extractor


class HipsNeckExtractor(Extractor):
    """shift = hips point, scale = |neck - hips| (reference hips_neck_extractor.py:6-13)."""
    kind = 'hips_neck'
