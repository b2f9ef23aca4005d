"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's validation metrics.

Follows metrics/mpjpe.py:29-45 (MPJPE), metrics/mrpe.py:38-76 (MRPE, with utils/world.py:16-63 for the cumulative world
location and hips_neck_extractor.py:6-13 for the hips point), metrics/pck.py:55-98 (PCK, with utils/tensors.py:12-40).
Each function returns the (sum, count) pair one ``update`` adds to the metric state; ``compute`` = 1000 * sum / count for
the two position errors (millimetres) and correct / total for PCK. Pinned by tests/golden/metrics.npz (the reference's own
classes run on two batches, tests/golden/make_golden.py section 9).
"""
from typing import Optional, Sequence, Tuple

import torch

from oracle import pose_head as O


def mpjpe_update(pred: torch.Tensor, gt: torch.Tensor, out_idx=None, in_idx=None) -> Tuple[torch.Tensor, int]:
    p = pred if out_idx is None else pred[:, :, list(out_idx)]
    g = gt if in_idx is None else gt[:, :, list(in_idx)]
    per_clip = torch.linalg.norm(p - g, dim=-1, ord=2).mean(dim=(-2, -1))            # mpjpe.py:40-41
    return per_clip.sum(), per_clip.numel()


def world_loc_from_changes(changes: torch.Tensor) -> torch.Tensor:
    """calculate_world_from_changes (world.py:47-63) for locations with a zero initial location: running sum."""
    return torch.cumsum(changes, dim=1)


def mrpe_update(pred: torch.Tensor, gt: torch.Tensor, world_pred: torch.Tensor, world_gt: torch.Tensor,
                pred_hips: Sequence[int] = (O.HIPS,), gt_hips: Sequence[int] = (O.HIPS,)) -> Tuple[torch.Tensor, int]:
    hp = pred[..., list(pred_hips), :].mean(dim=-2)                                   # mrpe.py:58-59
    hg = gt[..., list(gt_hips), :].mean(dim=-2)
    per_clip = torch.linalg.norm((world_pred + hp) - (world_gt + hg), dim=-1, ord=2).mean(dim=-1)   # :61-70
    return per_clip.sum(), per_clip.numel()


def pck_update(pred: torch.Tensor, gt: torch.Tensor, out_idx=None, in_idx=None, hips_col: Optional[int] = O.HIPS,
               mask_missing_joints: bool = True, mask_src: Optional[torch.Tensor] = None, norm: str = 'bbox',
               hips: Sequence[int] = (O.HIPS,), neck: Sequence[int] = (O.NECK,), threshold: float = 0.05,
               near_zero: float = 1e-5) -> Tuple[torch.Tensor, torch.Tensor]:
    """pred (B,T,Jp,2), gt (B,T,Jg,2); ``hips_col`` = position of the gt skeleton's hips joint in the common joint list."""
    src = gt if mask_src is None else mask_src
    g = gt if in_idx is None else gt[:, :, list(in_idx)]
    p = pred if out_idx is None else pred[:, :, list(out_idx)]
    if mask_missing_joints:                                                           # pck.py:67-70, tensors.py:29-40
        m = src if in_idx is None else src[:, :, list(in_idx)]
        mask = torch.all(m != 0, dim=-1)
        if hips_col is not None:
            mask = mask.clone()
            mask[..., hips_col] = True
    else:
        mask = torch.ones(g.shape[:-1], dtype=torch.bool)
    if norm == 'bbox':                                                                # pck.py:59-64
        boxes = O.get_bboxes(gt, near_zero)
        normalize = torch.linalg.norm(boxes[..., 1, :] - boxes[..., 0, :], dim=-1, ord=2)
    else:                                                                             # pck.py:55-57
        h, k = gt[..., list(hips), :].mean(dim=-2), gt[..., list(neck), :].mean(dim=-2)
        normalize = torch.linalg.norm(k - h, dim=-1, ord=2)
    mask = mask & ~(normalize < near_zero)[..., None]                                 # pck.py:82
    normalize = torch.where(normalize < near_zero, torch.ones_like(normalize), normalize)
    dist = torch.linalg.norm((p - g) / normalize[..., None, None], dim=-1, ord=2)    # pck.py:87-88
    return (dist[mask] < threshold).sum(), mask.sum()
