from .pose_former import PoseFormer
