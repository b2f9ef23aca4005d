"""Which kernel sources a profiles/traffic.json entry was measured on: entry name -> sha256[:16] over those files. bench.py prints a
stored traffic figure only while the stamp still matches the sources in the tree (a changed kernel drops its entries instead of
carrying a stale number); tools/make_traffic.py writes the stamp."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pedestrians_video_2_carla_amd', 'csrc')
_HEAD = ['p2c_pose_head.hip', 'p2c_pose_head_dev.h', 'p2c_pose_head_pk.inc']
_CHAIN = ['p2c_pose_head_chain.hip', 'p2c_pose_head_chain_dev.h', 'p2c_pose_head_dev.h']
_MLP = ['p2c_mlp.hip', 'p2c_mlp_dev.h', 'p2c_adam_math.h']
_TRAIN = ['p2c_train.hip', 'p2c_train_stream.hip', 'p2c_train_dev.h', 'p2c_mlp_dev.h', 'p2c_pose_head_dev.h', 'p2c_pose_head_chain_dev.h',
          'p2c_adam_math.h']
SOURCES = (('pose_head_chain', _CHAIN), ('pose_head', _HEAD), ('mlp_', _MLP), ('adamw', ['p2c_optim.hip', 'p2c_adam_math.h']),
           ('train_', _TRAIN))


def src_sha16(entry_name: str):
    """sha256[:16] of the sources behind the kernel group `entry_name` ('group@B...' or 'group'); None for an unknown group."""
    group = entry_name.split('@')[0]
    for prefix, files in SOURCES:
        if group.startswith(prefix):
            h = hashlib.sha256()
            for f in files:
                with open(os.path.join(CSRC, f), 'rb') as fh:
                    h.update(fh.read())
            return h.hexdigest()[:16]
    return None
