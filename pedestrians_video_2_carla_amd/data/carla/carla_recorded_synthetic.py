"""Synthetic CarlaRecorded-shaped batches, generated on device.

Shapes / keys follow ``CarlaRecordedDataset._get_targets`` (reference data/carla/datasets/carla_recorded_dataset.py:8-25)
+ ``Projection2DMixin.process_projection_2d`` (mixins/dataset/projection_2d_mixin.py:209-232); the motion recipe is the
reference's own synthetic generator ``Carla2D3DIterableDataset.generate_batch`` (carla_2d3d_dataset.py:145-210): per
frame 3 random joints are rotated by U(-5 deg, +5 deg) per Euler axis, (age, gender) uniform over the four reference
skeletons, no world motion; targets are produced by pushing those pose changes through the projection layer -- here
the HIP pose head -- and the data-module normaliser. No network, no files: data = "synthetic".
"""
import math
from typing import Dict, List, Tuple

import torch

from pedestrians_video_2_carla_amd import ops
from pedestrians_video_2_carla_amd.data.base.base_datamodule import BaseDataModule
from pedestrians_video_2_carla_amd.data.base.base_transforms import BaseTransforms
from pedestrians_video_2_carla_amd.data.carla.reference import CARLA_REFERENCE_SKELETON_TYPES
from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
from pedestrians_video_2_carla_amd.transforms.rotation_conversions import euler_angles_to_matrix

Batch = Tuple[torch.Tensor, Dict[str, torch.Tensor], Dict[str, List]]


class SyntheticCarlaRecordedDataModule(BaseDataModule):
    def __init__(self, clip_length: int = 16, batch_size: int = 256, seed: int = 22742,
                 missing_joint_probabilities: float = 0.0, random_changes_each_frame: int = 3,
                 max_change_in_deg: float = 5.0, transform=BaseTransforms.hips_neck_bbox, **kwargs):
        super().__init__(data_nodes=CARLA_SKELETON, clip_length=clip_length, batch_size=batch_size,
                         transform=transform, **kwargs)
        self.seed = seed
        self.missing_joint_probabilities = missing_joint_probabilities
        self.random_changes_each_frame = random_changes_each_frame
        self.max_change_in_rad = math.radians(max_change_in_deg)

    def generate_batch(self, device, batch_size: int = None, seed_offset: int = 0) -> Batch:
        B, T, J = batch_size or self.batch_size, self.clip_length, len(CARLA_SKELETON)
        gen = torch.Generator().manual_seed(self.seed + seed_offset)        # host generator: rank-reproducible
        skel_type = torch.randint(0, 4, (B,), generator=gen)
        k = self.random_changes_each_frame
        pick = torch.rand(B, T, J, generator=gen).argsort(-1)[..., :k]
        val = (torch.rand(B, T, k, 3, generator=gen) * 2 - 1) * self.max_change_in_rad
        angles = torch.zeros(B, T, J, 3).scatter_(2, pick[..., None].expand(B, T, k, 3), val)
        miss = (torch.rand(B, T, J, generator=gen) < self.missing_joint_probabilities) \
            if self.missing_joint_probabilities > 0 else None

        pose_changes = euler_angles_to_matrix(angles.to(device), 'XYZ')
        st = skel_type.to(device=device, dtype=torch.int32)
        spec = ops.PoseHeadSpec(kind='pose_changes', transform='none')
        with torch.no_grad():
            _, o = ops.pose_head(pose_changes, spec, st, want=('projection_2d', 'absolute_pose_loc', 'absolute_pose_rot',
                                                               'relative_pose_loc', 'relative_pose_rot'))
            projection_2d = o['projection_2d'][..., :2].contiguous()
            targets = {
                'projection_2d': projection_2d,
                'absolute_pose_loc': o['absolute_pose_loc'], 'absolute_pose_rot': o['absolute_pose_rot'],
                'relative_pose_loc': o['relative_pose_loc'], 'relative_pose_rot': o['relative_pose_rot'],
                'world_loc': torch.zeros(B, T, 3, device=device),
                'world_rot': torch.eye(3, device=device).expand(B, T, 3, 3).contiguous(),
            }
            frames = projection_2d
            if self.transform_callable is not None:
                transformed = self.transform_callable(projection_2d)
                targets['projection_2d_transformed'] = transformed
                targets['projection_2d_shift'] = self.transform_callable.shift
                targets['projection_2d_scale'] = self.transform_callable.scale
                frames = transformed
            frames = frames.clone()
            if miss is not None:     # deformation hits the model input only (projection_2d_mixin.py:137-171,217-222)
                frames[miss.to(device)] = 0.0
        meta = {
            'age': [CARLA_REFERENCE_SKELETON_TYPES[i][0] for i in skel_type.tolist()],
            'gender': [CARLA_REFERENCE_SKELETON_TYPES[i][1] for i in skel_type.tolist()],
            'skel_type': st,          # pre-resolved index tensor: lets on_batch_start skip the per-clip python zip
        }
        return frames, targets, meta

    def train_batches(self, device, steps: int, rank: int = 0):
        for i in range(steps):
            yield self.generate_batch(device, seed_offset=rank + 1000 * i)
