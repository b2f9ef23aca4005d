from .linear_ae import LinearAE
