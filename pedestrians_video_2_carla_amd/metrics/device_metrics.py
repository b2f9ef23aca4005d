"""MPJPE / MRPE / PCK on the device (reference metrics/mpjpe.py:9-49, mrpe.py:9-80, pck.py:12-102)."""
import ctypes
from typing import Dict, Optional, Sequence, Type

import torch
import torch.distributed as dist

from pedestrians_video_2_carla_amd import _lib
from pedestrians_video_2_carla_amd.data.base.skeleton import Skeleton, get_common_indices
from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON


def _as_list(point) -> Sequence[int]:
    return [p.value for p in point] if isinstance(point, (list, tuple)) else [point.value]


def _iarr(values: Sequence[int]):
    return (ctypes.c_int32 * max(1, len(values)))(*values)


def _pairs(input_nodes, output_nodes, n_out: int, n_in: int):
    """(gmap[n_out]: gt joint of prediction joint j or -1, pmap[n_in]: prediction joint of gt joint i or -1)."""
    out_idx, in_idx = get_common_indices(input_nodes=input_nodes, output_nodes=output_nodes)
    outs = list(range(n_out)) if isinstance(out_idx, slice) else list(out_idx)
    ins = list(range(n_in)) if isinstance(in_idx, slice) else list(in_idx)
    gmap, pmap = [-1] * n_out, [-1] * n_in
    for o, i in zip(outs, ins):
        gmap[o], pmap[i] = i, o
    return gmap, pmap, ins


def _dev(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.P2CError(f'{what} must live on the GPU: the device metrics have no CPU fallback')
    return t.contiguous().float()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class DeviceMetric:
    """(sum, count) state of four doubles on the device; subclasses add to it in ``update``."""

    def __init__(self):
        self._state: Optional[torch.Tensor] = None

    def _ensure(self, device):
        if self._state is None or self._state.device != device:
            self._state = torch.zeros(4, dtype=torch.float64, device=device)
        return self._state

    def reset(self):
        if self._state is not None:
            self._state.zero_()

    def sync(self, group=None):
        if self._state is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self._state, op=dist.ReduceOp.SUM, group=group)

    def __call__(self, predictions, targets):
        self.update(predictions, targets)
        return self.compute()


class MPJPE(DeviceMetric):
    """Mean per-joint position error in millimetres over ``absolute_pose_loc`` (mpjpe.py:9-49)."""

    def __init__(self, input_nodes: Type[Skeleton] = CARLA_SKELETON, output_nodes: Type[Skeleton] = CARLA_SKELETON):
        super().__init__()
        self.input_nodes, self.output_nodes = input_nodes, output_nodes
        self._pred_hips = _as_list(output_nodes.get_hips_point())
        self._gt_hips = _as_list(input_nodes.get_hips_point())

    def _launch(self, predictions, targets, world_pred=None, world_gt=None):
        pred, gt = _dev(predictions['absolute_pose_loc'], 'predictions'), _dev(targets['absolute_pose_loc'], 'targets')
        B, T, Jp = pred.shape[:3]
        Jg = gt.shape[2]
        if Jp != 26:
            raise RuntimeError('predictions are expected on the 26-joint CARLA skeleton')
        assert pred.shape[:2] == gt.shape[:2]
        gmap, _, _ = _pairs(self.input_nodes, self.output_nodes, Jp, Jg)
        lib = _lib.lib()
        part = torch.empty(lib.p2c_eval_workspace_floats(B), dtype=torch.float32, device=pred.device)
        state = self._ensure(pred.device)
        with torch.cuda.device(pred.device):
            _lib.check(lib.p2c_eval_pose3d(pred.data_ptr(), gt.data_ptr(), B, T, Jg, _iarr(gmap), _iarr(self._pred_hips),
                                           len(self._pred_hips), _iarr(self._gt_hips), len(self._gt_hips),
                                           None if world_pred is None else world_pred.data_ptr(),
                                           None if world_gt is None else world_gt.data_ptr(), part.data_ptr(),
                                           state.data_ptr(), _stream()), 'p2c_eval_pose3d')

    def update(self, predictions: Dict[str, torch.Tensor], targets: Dict[str, torch.Tensor]):
        if 'absolute_pose_loc' not in predictions or 'absolute_pose_loc' not in targets:
            return                                    # mpjpe.py:43-44: missing keys are silently skipped
        if isinstance(self, MRPE):
            return self._update_mrpe(predictions, targets)
        # MPJPE proper: a separate state from MRPE's (slots 0, 1)
        self._launch(predictions, targets)

    def compute(self):
        s = self._state
        return 1000.0 * (s[0] / s[1]).float()


class MRPE(MPJPE):
    """Mean root (hips) position error in millimetres, hips + cumulative world location (mrpe.py:9-80)."""

    def _update_mrpe(self, predictions, targets):
        try:
            if predictions.get('world_loc_changes') is not None:
                wp = torch.cumsum(_dev(predictions['world_loc_changes'], 'world_loc_changes'), dim=1)   # world.py:47-63
            else:
                wp = _dev(predictions['world_loc'], 'world_loc')
            wg = torch.cumsum(_dev(targets['world_loc_changes'], 'world_loc_changes'), dim=1)
        except KeyError:
            return
        self._launch(predictions, targets, wp.contiguous(), wg.contiguous())

    def compute(self):
        s = self._state
        return 1000.0 * (s[2] / s[3]).float()


class PCK(DeviceMetric):
    """Percentage of correct keypoints (pck.py:12-102)."""

    def __init__(self, input_nodes: Type[Skeleton] = CARLA_SKELETON, output_nodes: Type[Skeleton] = CARLA_SKELETON,
                 mask_missing_joints: bool = True, key: str = 'projection_2d', threshold: float = 0.05,
                 get_normalization_tensor: Optional[str] = None):
        super().__init__()
        if callable(get_normalization_tensor):
            raise NotImplementedError('custom normalisation callables run on the host; use "hn" or the default bbox')
        self.input_nodes, self.output_nodes = input_nodes, output_nodes
        self.key, self.threshold, self.mask_missing_joints = key, threshold, mask_missing_joints
        self.norm_mode = 1 if get_normalization_tensor == 'hn' else 0
        hips = input_nodes.get_hips_point()
        self._hips_joint = -1 if isinstance(hips, (list, tuple)) else hips.value
        self._hips_idx, self._neck_idx = _as_list(hips), _as_list(input_nodes.get_neck_point())
        self.near_zero = 1e-5

    def update(self, predictions: Dict[str, torch.Tensor], targets: Dict[str, torch.Tensor]):
        if self.key not in predictions or self.key not in targets:
            return
        pred, gt = _dev(predictions[self.key], 'predictions'), _dev(targets[self.key], 'targets')
        B, T, Jp, Cp = pred.shape
        Jg, Cg = gt.shape[2], gt.shape[3]
        _, pmap, ins = _pairs(self.input_nodes, self.output_nodes, Jp, Jg)
        mask_missing = self.mask_missing_joints and 'projection_2d' in targets
        mask_src = _dev(targets['projection_2d'], 'targets') if mask_missing else None
        if mask_src is not None and mask_src.shape != gt.shape:
            raise RuntimeError('projection_2d and the evaluated key must have the same shape in the targets')
        hips_joint = self._hips_joint if self._hips_joint in ins else -1
        lib = _lib.lib()
        part = torch.empty(lib.p2c_eval_workspace_floats(B * T), dtype=torch.float32, device=pred.device)
        state = self._ensure(pred.device)
        with torch.cuda.device(pred.device):
            _lib.check(lib.p2c_eval_pck(pred.data_ptr(), gt.data_ptr(), None if mask_src is None else mask_src.data_ptr(),
                                        B * T, Jp, Cp, Jg, Cg, _iarr(pmap), int(mask_missing), hips_joint, self.norm_mode,
                                        _iarr(self._hips_idx), len(self._hips_idx), _iarr(self._neck_idx),
                                        len(self._neck_idx), self.threshold, self.near_zero, part.data_ptr(),
                                        state.data_ptr(), _stream()), 'p2c_eval_pck')

    def compute(self):
        s = self._state
        return (s[0] / s[2]).float()
