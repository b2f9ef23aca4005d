"""rot_3d: MSE between predicted and target absolute joint rotations (reference loss/rot_3d.py:9-37).

In training with a 6-D rotation output the value comes out of the fused pose head (``_fused``: the kernel reads the target
rotations and writes no rotation tensor, p2c_pose_head_desc.gt_rot). Otherwise the gradient reaches the network through
``absolute_pose_rot`` of the materialising pose head; its backward is the tangent-space HIP kernel
(``grad_absolute_pose_rot`` of ``p2c_pose_head_bwd``)."""
from typing import Dict, Type

from torch import Tensor
from torch.nn.modules import loss

from pedestrians_video_2_carla_amd.data.base.skeleton import Skeleton, get_common_indices


def calculate_loss_rot_3d(criterion: loss._Loss, input_nodes: Type[Skeleton], output_nodes: Type[Skeleton],
                          absolute_pose_rot: Tensor = None, targets: Dict[str, Tensor] = None, _fused=None, **kwargs) -> Tensor:
    if _fused is not None:
        value = _fused.get('rot_3d', input_nodes, output_nodes)
        if value is not None:
            return value
    if absolute_pose_rot is None or targets is None or 'absolute_pose_rot' not in targets:
        return None
    output_indices, input_indices = get_common_indices(input_nodes, output_nodes)
    return criterion(absolute_pose_rot[:, :, output_indices], targets['absolute_pose_rot'][:, :, input_indices])
