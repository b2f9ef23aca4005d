"""Output-type enums of the model plugins (names and values as in the reference, modules/flow/output_types.py:4-28:
they are part of checkpoints' hparams and of the CLI)."""
from enum import Enum

MovementsModelOutputType = Enum('MovementsModelOutputType', dict(
    pose_changes=0,        # per-frame rotation changes of every bone (default)
    absolute_loc_rot=1,    # absolute locations + rotations
    absolute_loc=2,        # absolute locations only (PoseFormer, Baseline3DPose)
    relative_rot=3,        # relative rotations, no accumulation over time
    pose_2d=4,             # 2-D pose -> 2-D pose (autoencoder flow)
))
TrajectoryModelOutputType = Enum('TrajectoryModelOutputType', dict(changes=0, loc_rot=1))
ClassificationModelOutputType = Enum('ClassificationModelOutputType', dict(multiclass=0, binary=1))
PoseEstimationModelOutputType = Enum('PoseEstimationModelOutputType', dict(heatmaps=100, pose_2d=4))
