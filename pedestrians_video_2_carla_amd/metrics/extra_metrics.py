"""The remaining validation metrics of the reference's flows, as tensor reductions that stay
on the batch's device: ``update`` adds into a small persistent state, nothing is read back until ``compute()``.

  MultiinputWrapper(MeanSquaredError)   metrics/multiinput_wrapper.py:9-69  (autoencoder flow's 'MSE', autoencoder.py:73-81)
  MissingJointsRatio                    metrics/missing_joints_ratio.py:9-77 (autoencoder flow's initial 'MJR', :63-71)
  FB_MPJPE / FB_WeightedMPJPE / FB_N_MPJPE / FB_MPJVE / FB_PA_MPJPE   metrics/fb/*.py (pose-lifting flow, pose_lifting.py:88-105)

The first two are pinned by tests/golden/metrics_extra.npz (the reference's own classes run on two batches). The FB_*
classes wrap ``third_party/video_pose_3d/common/loss.py`` (empty submodule in the reference checkout): the five functions
are restated from the published VideoPose3D definitions (Pavllo et al. 2019) and are PARITY-UNPINNED; the wrappers' own
quirks are kept -- MPJVE differentiates over the FLATTENED (clip x frame) axis, i.e. across clip boundaries
(fb_mpjve.py:26-31), and every batch is weighted by its frame count.
"""
from typing import Dict, Optional, Type

import torch
import torch.distributed as dist

from pedestrians_video_2_carla_amd.data.base.skeleton import Skeleton, get_common_indices
from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON


def get_missing_joints_mask(common_gt: torch.Tensor, hips=None, input_indices=None) -> torch.Tensor:
    """(..., K) bool: joints whose ground truth is not the 'perfect zero' of an undetected joint; the hips joint of the input
    skeleton -- when it is a single joint among the common ones -- always counts (reference utils/tensors.py:29-40)."""
    mask = (common_gt != 0).all(dim=-1)
    if hips is not None:
        col = hips.value if isinstance(input_indices, slice) else list(input_indices).index(hips.value)
        mask[..., col] = True
    return mask


class _StateMetric:
    """State = one float64 vector on the device of the first update (sum-reducible across ranks, like torchmetrics'
    ``dist_reduce_fx='sum'``)."""
    _n_state = 2

    def __init__(self):
        self._state: Optional[torch.Tensor] = None

    def _ensure(self, device):
        if self._state is None or self._state.device != device:
            self._state = torch.zeros(self._n_state, dtype=torch.float64, device=device)
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


class MeanSquaredError(_StateMetric):
    """torchmetrics.MeanSquaredError: sum of squared errors / number of elements."""

    def update(self, preds: torch.Tensor, target: torch.Tensor):
        st = self._ensure(preds.device)
        st[0] += ((preds - target) ** 2).sum().double()
        st[1] += target.numel()

    def compute(self):
        return (self._state[0] / self._state[1]).float()


class MultiinputWrapper(_StateMetric):
    """Pick ``pred_key`` / ``target_key`` out of the prediction / target dicts, gather the common joints, drop the joints
    whose ground truth is missing, feed the base metric (multiinput_wrapper.py:49-69)."""

    def __init__(self, base_metric, pred_key: str, target_key: str, input_nodes: Type[Skeleton], output_nodes: Type[Skeleton],
                 mask_missing_joints: bool = True, **kwargs):
        super().__init__()
        self.base_metric = base_metric
        self.pred_key, self.target_key = pred_key, target_key
        if input_nodes is None and output_nodes is None:       # not a per-joint metric
            self._input_indices = self._output_indices = self._input_hips = None
            self._mask_missing_joints = False
        else:
            self._output_indices, self._input_indices = get_common_indices(input_nodes, output_nodes)
            self._mask_missing_joints = mask_missing_joints
            self._input_hips = input_nodes.get_hips_point()
            if isinstance(self._input_hips, (list, tuple)):
                self._input_hips = None

    def update(self, predictions: Dict[str, torch.Tensor], targets: Dict[str, torch.Tensor]):
        if self._input_indices is None and self._output_indices is None:
            return self.base_metric.update(predictions[self.pred_key], torch.atleast_1d(targets[self.target_key]))
        common_pred = predictions[self.pred_key][..., self._output_indices, :]
        common_gt = targets[self.target_key][..., self._input_indices, :]
        if self._mask_missing_joints:
            # boolean-mask indexing has a data-dependent shape (a host sync): for the squared-error base metric the same sums
            # come from a masked reduction; any other base metric gets the reference's gathered tensors
            mask = get_missing_joints_mask(common_gt, self._input_hips, self._input_indices)
            if isinstance(self.base_metric, MeanSquaredError):
                st = self.base_metric._ensure(common_pred.device)
                st[0] += (((common_pred - common_gt) ** 2) * mask[..., None]).sum().double()
                st[1] += mask.sum().double() * common_gt.shape[-1]
                return None
            common_pred, common_gt = common_pred[mask], common_gt[mask]
        return self.base_metric.update(common_pred, common_gt)

    def compute(self):
        return self.base_metric.compute()

    def reset(self):
        self.base_metric.reset()

    def sync(self, group=None):
        self.base_metric.sync(group)

    @property
    def _state(self):
        return self.base_metric._state

    @_state.setter
    def _state(self, value):
        pass


class MissingJointsRatio(_StateMetric):
    """Share of predicted joints that are exactly zero in ``projection_2d`` (missing_joints_ratio.py:41-63)."""

    def __init__(self, input_nodes: Type[Skeleton] = CARLA_SKELETON, output_nodes: Type[Skeleton] = CARLA_SKELETON,
                 report_per_joint: bool = False, **kwargs):
        super().__init__()
        self.input_nodes, self.output_nodes, self.report_per_joint = input_nodes, output_nodes, report_per_joint
        self.output_indices, self.input_indices = get_common_indices(input_nodes, output_nodes)
        self.output_num_joints = len(range(len(output_nodes))[self.output_indices]) if isinstance(self.output_indices, slice) \
            else len(self.output_indices)
        self._n_state = self.output_num_joints + 1             # present joints per joint, frames seen

    def update(self, predictions: Dict[str, torch.Tensor], targets: Dict[str, torch.Tensor]):
        if 'projection_2d' not in predictions:                  # MJR only makes sense with 'absolute' predictions (:48-50)
            return
        prediction = predictions['projection_2d'][:, :, self.output_indices]
        st = self._ensure(prediction.device)
        st[:-1] += prediction.all(dim=-1).sum(dim=tuple(range(prediction.ndim - 2))).double()
        st[-1] += float(torch.Size(prediction.shape[:-2]).numel())

    def compute(self):
        present, total = self._state[:-1], self._state[-1]
        mean = (1.0 - present.sum() / (self.output_num_joints * total)).float()
        if not self.report_per_joint:
            return mean
        per_joint = (1.0 - present / total).float()
        idx = range(len(self.output_nodes)) if isinstance(self.output_indices, slice) else self.output_indices
        return {'mean': mean, 'per_joint': {self.output_nodes(i).name: per_joint[k] for k, i in enumerate(idx)}}


# ---- VideoPose3D common/loss.py, restated (parity-unpinned: the submodule is empty in the reference checkout) ---------------
def mpjpe(predicted: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Mean per-joint position error: mean Euclidean distance over every leading axis."""
    return torch.mean(torch.norm(predicted - target, dim=-1))


def weighted_mpjpe(predicted: torch.Tensor, target: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    return torch.mean(w * torch.norm(predicted - target, dim=-1))


def n_mpjpe(predicted: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Scale-normalised MPJPE: the prediction is rescaled per frame by <target, predicted> / <predicted, predicted>."""
    norm_predicted = torch.mean(torch.sum(predicted ** 2, dim=3, keepdim=True), dim=2, keepdim=True)
    norm_target = torch.mean(torch.sum(target * predicted, dim=3, keepdim=True), dim=2, keepdim=True)
    return mpjpe(norm_target / norm_predicted * predicted, target)


def mean_velocity_error(predicted: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Mean per-joint velocity error: first differences along axis 0."""
    return torch.mean(torch.norm(torch.diff(predicted, dim=0) - torch.diff(target, dim=0), dim=-1))


def p_mpjpe(predicted: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """MPJPE after a per-frame similarity (Procrustes) alignment of the prediction to the target; (N, J, 3) inputs.
    Batched 3x3 SVD on the tensors' device, in double (the published version runs numpy on the host)."""
    X, Y = target.double(), predicted.double()
    muX, muY = X.mean(dim=1, keepdim=True), Y.mean(dim=1, keepdim=True)
    X0, Y0 = X - muX, Y - muY
    normX = torch.sqrt((X0 ** 2).sum(dim=(1, 2), keepdim=True))
    normY = torch.sqrt((Y0 ** 2).sum(dim=(1, 2), keepdim=True))
    X0, Y0 = X0 / normX, Y0 / normY
    H = X0.transpose(1, 2) @ Y0
    U, s, Vt = torch.linalg.svd(H)
    V = Vt.transpose(1, 2)
    R = V @ U.transpose(1, 2)
    sign = torch.sign(torch.linalg.det(R))[:, None]             # no reflections
    V = torch.cat((V[:, :, :-1], V[:, :, -1:] * sign[:, None]), dim=2)
    s = torch.cat((s[:, :-1], s[:, -1:] * sign), dim=1)
    R = V @ U.transpose(1, 2)
    a = s.sum(dim=1, keepdim=True)[..., None] * normX / normY
    t = muX - a * (muY @ R)
    aligned = a * (Y @ R) + t
    return torch.mean(torch.norm(aligned - X, dim=-1))


class _FBMetric(_StateMetric):
    """errors += frames * metric(batch), total += frames; compute() = 1000 * errors / total, millimetres (fb_mpjpe.py:18-41)."""

    def _metric(self, prediction, target):
        raise NotImplementedError

    def update(self, predictions: Dict[str, torch.Tensor], targets: Dict[str, torch.Tensor]):
        if 'absolute_pose_loc' not in predictions or 'absolute_pose_loc' not in targets:
            return                                               # KeyError / AssertionError are swallowed in the reference
        prediction, target = predictions['absolute_pose_loc'], targets['absolute_pose_loc']
        if prediction.shape != target.shape:
            return
        frames = float(torch.Size(prediction.shape[:-2]).numel())
        st = self._ensure(prediction.device)
        st[0] += frames * self._metric(prediction, target).double()
        st[1] += frames

    def compute(self):
        return (1000.0 * self._state[0] / self._state[1]).float()


class FB_MPJPE(_FBMetric):
    def _metric(self, prediction, target):
        return mpjpe(prediction.reshape((-1,) + prediction.shape[-2:]), target.reshape((-1,) + target.shape[-2:]))


class FB_WeightedMPJPE(_FBMetric):
    def __init__(self, w: Optional[torch.Tensor] = None):
        super().__init__()
        self.w = w

    def _metric(self, prediction, target):
        w = self.w if self.w is not None else torch.ones((1, 1, prediction.shape[-2]))
        if w.shape[0] != torch.Size(prediction.shape[:-2]).numel():
            w = w.repeat((*prediction.shape[:-2], 1))           # fb_weighted_mpjpe.py:33-36 (shape quirk included)
        return weighted_mpjpe(prediction, target, w.to(prediction.device))


class FB_N_MPJPE(_FBMetric):
    def _metric(self, prediction, target):
        return n_mpjpe(prediction, target)                       # the 4-D (B, T, J, 3) tensors, as the reference passes them


class FB_MPJVE(_FBMetric):
    def _metric(self, prediction, target):
        return mean_velocity_error(prediction.reshape((-1,) + prediction.shape[-2:]), target.reshape((-1,) + target.shape[-2:]))


class FB_PA_MPJPE(_FBMetric):
    def _metric(self, prediction, target):
        return p_mpjpe(prediction.reshape((-1,) + prediction.shape[-2:]), target.reshape((-1,) + target.shape[-2:]))
