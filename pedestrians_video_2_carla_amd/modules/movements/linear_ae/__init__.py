from .linear_ae import LinearAE
from .linear_ae_residual import LinearAEResidual, LinearAEResidualLeaky
