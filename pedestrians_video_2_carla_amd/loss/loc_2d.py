"""loc_2d: masked MSE between the (transformed) 2-D projection and its target (reference loss/loc_2d.py:69-89)."""
from typing import Dict

from torch import Tensor

from pedestrians_video_2_carla_amd.loss.base_pose_loss import BasePoseLoss


class Loc2DPoseLoss(BasePoseLoss):
    fused_name = 'loc_2d'

    def _extract_gt_targets(self, targets: Dict[str, Tensor], **kwargs) -> Tensor:
        assert 'projection_2d' in targets or 'projection_2d_transformed' in targets, \
            'Either projection_2d or projection_2d_transformed must be provided.'
        return targets['projection_2d_transformed'] if 'projection_2d_transformed' in targets else targets['projection_2d']

    def _extract_predicted_targets(self, projection_2d: Tensor = None, projection_2d_transformed: Tensor = None,
                                   **kwargs) -> Tensor:
        assert projection_2d is not None or projection_2d_transformed is not None, \
            'Either projection_2d or projection_2d_transformed must be provided.'
        return projection_2d_transformed if projection_2d_transformed is not None else projection_2d
