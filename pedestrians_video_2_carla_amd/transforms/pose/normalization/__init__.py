from .extractor import Extractor
from .normalizer import Normalizer
from .denormalizer import DeNormalizer
