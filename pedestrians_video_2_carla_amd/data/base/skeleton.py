"""Skeleton registry and input<->output joint index mapping.

Mirrors the reference's ``data/base/skeleton.py`` (register_skeleton :19, get_common_indices :26-56)
and the external ``pedestrians_scenarios.karma.pose.skeleton.Skeleton`` base enum (not in the reference
tree; contract restated from its call sites: ``get_hips_point``/``get_neck_point`` used by
transforms/pose/normalization/hips_neck_extractor.py:6-13, ``.value`` used everywhere as tensor index).
"""
from enum import Enum
from functools import lru_cache
from typing import Type

SKELETONS = {}
MAPPINGS = {}


class Skeleton(Enum):
    """Base of every skeleton enum: member value == joint index in tensors."""

    @classmethod
    def get_hips_point(cls):
        raise NotImplementedError()

    @classmethod
    def get_neck_point(cls):
        raise NotImplementedError()

    @classmethod
    def get_flip_mask(cls):
        raise NotImplementedError()

    @classmethod
    def get_edges(cls):
        raise NotImplementedError()

    @classmethod
    def get_colors(cls):
        return {k: (255, 255, 255, 255) for k in cls}

    @classmethod
    def get_edge_index(cls):
        import torch
        edges = cls.get_edges()
        fwd = [(a.value, b.value) for a, b in edges]
        bwd = [(b, a) for a, b in fwd]
        return torch.tensor(fwd + bwd, dtype=torch.long).t().contiguous()


def get_skeleton_type_by_name(name):
    return SKELETONS[name]


def get_skeleton_name_by_type(skeleton):
    return skeleton.__name__


def register_skeleton(name, skeleton, mapping=None):
    """``mapping`` = list of (CARLA_SKELETON member, this-skeleton member) pairs."""
    SKELETONS[name] = skeleton
    if mapping is not None:
        MAPPINGS[skeleton] = mapping


@lru_cache(maxsize=None)
def get_common_indices(input_nodes: Type[Skeleton] = None, output_nodes: Type[Skeleton] = None):
    """(output_indices, input_indices) of the joints both skeletons share, in matching order.

    Same skeletons (or an unmapped one) -> two full slices. Otherwise both lists are ordered by the
    CARLA index the joints map to, so ``pred[..., out_idx, :]`` and ``gt[..., in_idx, :]`` line up
    (reference data/base/skeleton.py:26-56).
    """
    if (input_nodes == output_nodes) \
            or (input_nodes is not None and input_nodes not in MAPPINGS) \
            or (output_nodes is not None and output_nodes not in MAPPINGS):
        return slice(None), slice(None)

    def carla_pairs(nodes):
        return [(c.value, o.value) for (c, o) in MAPPINGS[nodes]]

    if output_nodes is None:
        pairs = carla_pairs(input_nodes)
        return tuple(c for c, _ in pairs), tuple(i for _, i in pairs)
    if input_nodes is None:
        pairs = carla_pairs(output_nodes)
        return tuple(o for _, o in pairs), tuple(c for c, _ in pairs)

    inp = dict(carla_pairs(input_nodes))
    out = dict(carla_pairs(output_nodes))
    common = sorted(set(inp) & set(out))
    return [out[c] for c in common], [inp[c] for c in common]
