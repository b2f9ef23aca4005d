"""CARLA walker skeleton: 26 bones, enum value == DFS position in the bone tree.

The enum itself is external to the reference (``pedestrians_scenarios.karma.pose.skeleton.CARLA_SKELETON``,
imported at reference data/carla/skeleton.py:1). Member order is reconstructed from the bone tree
``data/carla/files/structure.yaml:1-27`` (== key order of every ``sk_*.yaml``); P3dPose indexes tensors by that
DFS order (walker_control/p3d_pose.py:126) and HipsNeckExtractor(CARLA_SKELETON) indexes the same tensors by
``.value`` (reference_skeletons_denormalizer.py:37), so values must equal DFS positions. SURVEY.md appendix A.1.
"""
from pedestrians_video_2_carla_amd.data.base.skeleton import Skeleton, register_skeleton


class CARLA_SKELETON(Skeleton):
    crl_root = 0
    crl_hips__C = 1
    crl_spine__C = 2
    crl_spine01__C = 3
    crl_shoulder__L = 4
    crl_arm__L = 5
    crl_foreArm__L = 6
    crl_hand__L = 7
    crl_neck__C = 8
    crl_Head__C = 9
    crl_eye__L = 10
    crl_eye__R = 11
    crl_shoulder__R = 12
    crl_arm__R = 13
    crl_foreArm__R = 14
    crl_hand__R = 15
    crl_thigh__R = 16
    crl_leg__R = 17
    crl_foot__R = 18
    crl_toe__R = 19
    crl_toeEnd__R = 20
    crl_thigh__L = 21
    crl_leg__L = 22
    crl_foot__L = 23
    crl_toe__L = 24
    crl_toeEnd__L = 25

    @classmethod
    def get_hips_point(cls):
        return CARLA_SKELETON.crl_hips__C

    @classmethod
    def get_neck_point(cls):
        return CARLA_SKELETON.crl_neck__C

    @classmethod
    def get_flip_mask(cls):
        swap = {}
        for m in cls:
            if m.name.endswith('__L'):
                swap[m.value] = cls[m.name[:-1] + 'R'].value
            elif m.name.endswith('__R'):
                swap[m.value] = cls[m.name[:-1] + 'L'].value
            else:
                swap[m.value] = m.value
        return tuple(swap[i] for i in range(len(cls)))

    @classmethod
    def get_edges(cls):
        members = list(cls)
        return [(members[p], members[c]) for c, p in enumerate(PARENTS) if p >= 0]


# parent joint index per joint (root = -1); depth <= 7 edges (root..hand)
PARENTS = (-1, 0, 1, 2, 3, 4, 5, 6, 3, 8, 9, 9, 3, 12, 13, 14, 1, 16, 17, 18, 19, 1, 21, 22, 23, 24)

register_skeleton('CARLA_SKELETON', CARLA_SKELETON, [(k, k) for k in CARLA_SKELETON])
