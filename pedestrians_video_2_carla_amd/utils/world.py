"""World transform helpers (reference utils/world.py:6-63).

The per-frame scan ``rot[t] = rot[t-1] @ drot[t]`` runs inside the HIP pose head (p2c_pose_head.hip ``world_step``);
these helpers only build the constant tensors the API hands around.
"""
from typing import Tuple

import torch
from torch import Tensor


def zero_world_loc(shape: Tuple, device: torch.device) -> Tensor:
    return torch.zeros((*shape, 3), device=device)


def zero_world_rot(shape: Tuple, device: torch.device) -> Tensor:
    return torch.eye(3, device=device).expand(*shape, 3, 3).contiguous()


def calculate_world_from_changes(shape: Tuple, device: torch.device, world_loc_change_batch: Tensor = None,
                                 world_rot_change_batch: Tensor = None, initial_world_loc: Tensor = None,
                                 initial_world_rot: Tensor = None) -> Tuple[Tensor, Tensor]:
    """API-compatible helper (reference utils/world.py:16-63) for callers OUTSIDE the fused path -- e.g. the optional
    ``targets['world_*_changes']`` of pose_lifting.py:186-194, which CarlaRecorded batches do not carry.
    rot[t] = rot[t-1] @ drot[t] is a prefix product; it is evaluated as a log-depth scan of batched matmuls."""
    batch_size, clip_length, *_ = shape
    if initial_world_loc is None:
        initial_world_loc = zero_world_loc((batch_size,), device)
    if initial_world_rot is None:
        initial_world_rot = zero_world_rot((batch_size,), device)
    if world_loc_change_batch is None and world_rot_change_batch is None:
        return (initial_world_loc.unsqueeze(1).repeat(1, clip_length, 1),
                initial_world_rot.unsqueeze(1).repeat(1, clip_length, 1, 1))
    if world_loc_change_batch is None:
        world_loc_change_batch = zero_world_loc((batch_size, clip_length), device)
    if world_rot_change_batch is None:
        world_rot_change_batch = zero_world_rot((batch_size, clip_length), device)
    world_loc = initial_world_loc.unsqueeze(1) + torch.cumsum(world_loc_change_batch, dim=1)
    prod = world_rot_change_batch.clone()
    step = 1
    while step < clip_length:       # Hillis-Steele inclusive scan: P[t] = P[t - step] @ P[t]
        prod = torch.cat((prod[:, :step], prod[:, :-step] @ prod[:, step:]), dim=1)
        step *= 2
    return world_loc, initial_world_rot.unsqueeze(1) @ prod
