"""Validation metrics computed on the device (reference package ``metrics/``).

Same ``update(predictions, targets)`` / ``compute()`` / ``reset()`` contract as the reference's torchmetrics objects, but
every update is one streaming HIP launch (+ a one-workgroup accumulate) into a persistent device state; nothing syncs
with the host until ``compute()``. ``sync()`` all-reduces the state for multi-GPU validation (torchmetrics'
``dist_reduce_fx='sum'``)."""
from .device_metrics import MPJPE, MRPE, PCK, DeviceMetric
from .extra_metrics import (FB_MPJPE, FB_MPJVE, FB_N_MPJPE, FB_PA_MPJPE, FB_WeightedMPJPE, MeanSquaredError,
                            MissingJointsRatio, MultiinputWrapper)

__all__ = ['DeviceMetric', 'MPJPE', 'MRPE', 'PCK', 'MultiinputWrapper', 'MeanSquaredError', 'MissingJointsRatio', 'FB_MPJPE',
           'FB_WeightedMPJPE', 'FB_N_MPJPE', 'FB_MPJVE', 'FB_PA_MPJPE']
