"""per_joint_loc_2d: loc_2d with one weight per joint (reference loss/per_joint_loc_2d.py:8-26; ``--loss_params``).

``sum(mask * w_j * K * (pred - gt)^2) / numel(gt[mask])`` over the K common joints. Cold path (SURVEY.md section 8f
rank 2): device-agnostic tensor ops without boolean-mask gathers (no host sync); gradients by autograd."""
from typing import Iterable

import torch
from torch import Tensor

from pedestrians_video_2_carla_amd.loss.base_pose_loss import index_list
from pedestrians_video_2_carla_amd.loss.loc_2d import Loc2DPoseLoss


class PerJointLoc2DPoseLoss(Loc2DPoseLoss):
    fused_name = None          # never served by the fused pose head: the weights are not part of its loss

    def __init__(self, loss_params: Iterable, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._weights = torch.tensor(list(loss_params), dtype=torch.float32)

    def __call__(self, **kwargs) -> Tensor:
        gt = self._extract_gt_targets(**kwargs)[..., 0:2]
        pred = self._extract_predicted_targets(**kwargs)[..., 0:2]
        n_common = min(pred.shape[-2], gt.shape[-2])
        oi, ii = index_list(self._output_indices, n_common), index_list(self._input_indices, n_common)
        common_pred, common_gt = pred[..., oi, :], gt[..., ii, :]
        weights = self._weights.to(common_gt.device)[ii].unsqueeze(-1)
        weights = weights * len(weights)
        sq = weights * (common_pred - common_gt) ** 2
        if not self._mask_missing_joints:
            return sq.sum() / common_gt.numel()
        mask = torch.all(common_gt != 0, dim=-1)                 # utils/tensors.py:29-40
        hips = self.hips_column(gt.shape[-2])
        if hips >= 0:
            mask = mask.clone()
            mask[..., hips] = True
        m = mask.unsqueeze(-1).to(sq.dtype)
        return (m * sq).sum() / (mask.sum() * common_gt.shape[-1])
