"""rot_3d: MSE between predicted and target absolute joint rotations (reference loss/rot_3d.py:9-37).

The gradient reaches the network through ``absolute_pose_rot`` of the materialising pose head; its backward is the
tangent-space HIP kernel (``grad_absolute_pose_rot`` of ``p2c_pose_head_bwd``)."""
from typing import Dict, Type

from torch import Tensor
from torch.nn.modules import loss

from pedestrians_video_2_carla_amd.data.base.skeleton import Skeleton, get_common_indices


def calculate_loss_rot_3d(criterion: loss._Loss, input_nodes: Type[Skeleton], output_nodes: Type[Skeleton],
                          absolute_pose_rot: Tensor = None, targets: Dict[str, Tensor] = None, **kwargs) -> Tensor:
    if absolute_pose_rot is None or targets is None or 'absolute_pose_rot' not in targets:
        return None
    output_indices, input_indices = get_common_indices(input_nodes, output_nodes)
    return criterion(absolute_pose_rot[:, :, output_indices], targets['absolute_pose_rot'][:, :, input_indices])
