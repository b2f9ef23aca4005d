"""ZeroTrajectory: the pedestrian never moves in the world (reference modules/trajectory/zero.py:5-20).

``forward`` keeps the reference contract (zeros (B,T,3), identity (B,T,3,3)). ``is_identity`` lets the HIP projection
layer constant-fold the world transform instead of reading 48 bytes per frame of zeros and ones.
"""
import torch

from pedestrians_video_2_carla_amd.modules.trajectory.trajectory import TrajectoryModel


class ZeroTrajectory(TrajectoryModel):
    is_identity = True

    def forward(self, x, *args, **kwargs):
        lead = tuple(x.shape[:2])
        return (torch.zeros(lead + (3,), device=x.device),
                torch.eye(3, device=x.device).expand(lead + (3, 3)).contiguous())

    def configure_optimizers(self):
        return {}
