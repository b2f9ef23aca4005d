from .extractor import Extractor


class HipsNeckBBoxFallbackExtractor(Extractor):
    """hips-neck, with scale <- 0.5748 * bbox half-height in frames whose hips or neck point is missing
    (reference hips_neck_bbox_fallback_extractor.py:9-40; the shift fallback there writes into a temporary and has no
    effect -- kept that way)."""
    kind = 'hips_neck_bbox'
