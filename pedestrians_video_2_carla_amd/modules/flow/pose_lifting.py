"""``LitPoseLiftingFlow``: 2-D keypoints -> model -> (HIP pose head: FK, projection, transform, losses).

Mirrors reference modules/flow/pose_lifting.py:25-195. ``_inner_step`` keeps the reference order
(movements model -> trajectory model -> projection layer -> eval_slice -> transform_callable) but, when every requested
loss is one the HIP pose head computes (loc_2d / loc_3d / loc_2d_3d) and the data module's transform is one of the
built-in normalisers, the projection layer, the transform and the losses are ONE kernel launch forward and ONE backward
(``ProjectionModule.fused_losses``). Otherwise the materialising ``ProjectionModule.forward`` feeds the generic loss
registry exactly like the reference does.

``lean_train_outputs`` (default True): in ``training_step`` the detached tensors the reference returns for logging
(``preds`` values) are not materialised -- they are ``None``, which ``_get_outputs`` already allows (base.py:430-433).
Validation / test / predict always materialise everything.
"""
from typing import Dict

import torch

from pedestrians_video_2_carla_amd.data.base.skeleton import get_common_indices
from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
from pedestrians_video_2_carla_amd.loss.fused import FusedLosses
from pedestrians_video_2_carla_amd.modules.flow.base import LitBaseFlow
from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType, TrajectoryModelOutputType
from pedestrians_video_2_carla_amd.modules.layers.projection import ProjectionModule
from pedestrians_video_2_carla_amd.modules.movements.linear_ae import LinearAE, LinearAEResidual, LinearAEResidualLeaky
from pedestrians_video_2_carla_amd.modules.movements.seq2seq import (Seq2Seq, Seq2SeqEmbeddings, Seq2SeqResidualA, Seq2SeqResidualB,
                                                                   Seq2SeqResidualC)
from pedestrians_video_2_carla_amd.modules.movements.pose_former import PoseFormer
from pedestrians_video_2_carla_amd.modules.movements.zero import ZeroMovements
from pedestrians_video_2_carla_amd.modules.trajectory.zero import ZeroTrajectory
from pedestrians_video_2_carla_amd.ops import joint_maps
from pedestrians_video_2_carla_amd.utils.world import calculate_world_from_changes

_FUSABLE_LOSSES = {'loc_2d', 'loc_3d', 'loc_2d_3d'}
# the rotation losses ride in the same launches when the network emits 6-D rotations
_FUSABLE_ROT_LOSSES = {'rot_3d', 'loc_rot_3d', 'loc_2d_loc_rot_3d', 'weighted_loc_2d_loc_rot_3d'}
_PROJECTION_KEYS = ('relative_pose_loc', 'relative_pose_rot', 'absolute_pose_loc', 'absolute_pose_rot', 'world_loc',
                    'world_rot')


class LitPoseLiftingFlow(LitBaseFlow):
    def __init__(self, *args, lean_train_outputs: bool = True, **kwargs):
        super().__init__(*args, **kwargs)
        self.lean_train_outputs = lean_train_outputs
        self.projection = ProjectionModule(
            movements_output_type=self.movements_model.output_type,
            trajectory_output_type=self.trajectory_model.output_type,
        )
        # let the pose head orthonormalise the 6-D rotations itself (one 24 B read instead of 36 B write + read)
        if hasattr(self.movements_model, 'rotation_output_format') and self.movements_model.output_type in (
                MovementsModelOutputType.pose_changes, MovementsModelOutputType.relative_rot):
            self.movements_model.rotation_output_format = 'rotation_6d'
        self._datamodule = None

    @classmethod
    def get_available_models(cls) -> Dict[str, Dict[str, torch.nn.Module]]:
        return {
            'movements': {m.__name__: m for m in [ZeroMovements, LinearAE, Seq2Seq, Seq2SeqEmbeddings, Seq2SeqResidualA,
                                                  Seq2SeqResidualB, Seq2SeqResidualC, LinearAEResidual,
                                                  LinearAEResidualLeaky, PoseFormer]},
            'trajectory': {m.__name__: m for m in [ZeroTrajectory]},
        }

    @classmethod
    def get_default_models(cls) -> Dict[str, torch.nn.Module]:
        return {'trajectory': ZeroTrajectory, 'movements': LinearAE}

    def get_metrics(self):
        """reference pose_lifting.py:88-105: MPJPE, MRPE (HIP reductions) and the five FB_* wrappers (tensor reductions on
        the device; VideoPose3D definitions restated, parity-unpinned -- metrics/extra_metrics.py)."""
        from pedestrians_video_2_carla_amd.metrics import (FB_MPJPE, FB_MPJVE, FB_N_MPJPE, FB_PA_MPJPE, FB_WeightedMPJPE, MPJPE,
                                                         MRPE)
        nodes = dict(input_nodes=self.movements_model.input_nodes, output_nodes=self.movements_model.output_nodes)
        metrics = {'MPJPE': MPJPE(**nodes), 'MRPE': MRPE(**nodes)}
        if self.movements_model.input_nodes is self.movements_model.output_nodes:     # the FB wrappers compare whole tensors
            metrics.update({'FB_MPJPE': FB_MPJPE(), 'FB_WeightedMPJPE': FB_WeightedMPJPE(), 'FB_PA_MPJPE': FB_PA_MPJPE(),
                            'FB_N_MPJPE': FB_N_MPJPE(), 'FB_MPJVE': FB_MPJVE()})
        return metrics

    def _get_crucial_keys(self):
        return [self._outputs_key, *_PROJECTION_KEYS]

    # ---- data module seam (pose_lifting.py:167-170 reads self.trainer.datamodule) --------------------------------
    def attach_datamodule(self, datamodule):
        self._datamodule = datamodule

    @property
    def datamodule(self):
        dm = self._datamodule
        if dm is None and getattr(self, 'trainer', None) is not None:
            dm = getattr(self.trainer, 'datamodule', None)
        if dm is None:
            raise RuntimeError('LitPoseLiftingFlow needs trainer.datamodule (its transform_callable is applied to the '
                               'projection); attach one with Trainer.fit(...) or flow.attach_datamodule(dm)')
        return dm

    def _on_batch_start(self, batch, batch_idx):
        self.projection.on_batch_start(batch, batch_idx)
        # per-batch state of the two-launch train step: the number of unmasked 2-D target pairs is a property of the targets
        # alone (the reference's on_batch_start is where per-batch constants are derived too, projection.py:52-71)
        self._pair_counts = None
        plan = self._fused_train_plan(batch[0], batch[1]) if self.training else None
        if plan is not None:
            from pedestrians_video_2_carla_amd import ops
            spec, gt2d, _gt3d = plan
            # one persistent buffer and one prebuilt launch descriptor: a captured step reads this address for every batch
            # staged later, and staging a batch costs one ctypes call here
            n, dev = len(batch[0]), batch[0].device
            buf = getattr(self, '_pair_counts_buf', None)
            if buf is None or buf.shape[0] != n or buf.device != dev:
                buf = self._pair_counts_buf = torch.zeros(n, dtype=torch.float32, device=dev)
                self._pair_counter = None
            if gt2d is None:
                self._pair_counts = buf.zero_()
            else:
                pc = getattr(self, '_pair_counter', None)
                if pc is None or pc.spec is not spec or not pc.matches(gt2d.shape, gt2d.device):
                    pc = self._pair_counter = ops.PairCounter(spec, gt2d, out=buf)
                    pc.spec = spec
                self._pair_counts = pc(gt2d if gt2d.is_contiguous() else gt2d.contiguous())

    # ---- fused-path configuration -----------------------------------------------------------------------------------
    def _fusable(self, transform_callable) -> bool:
        if self.movements_model.output_nodes is not CARLA_SKELETON:
            return False
        names = {name for (name, *_rest) in self._losses_to_calculate}
        if names - _FUSABLE_LOSSES - _FUSABLE_ROT_LOSSES:
            return False
        if names & _FUSABLE_ROT_LOSSES and not self._six_d_rotations():
            return False
        return transform_callable is None or getattr(transform_callable, 'kind', None) is not None

    def _six_d_rotations(self) -> bool:
        model = self.movements_model
        return (getattr(model, 'output_type', None) in (MovementsModelOutputType.pose_changes, MovementsModelOutputType.relative_rot)
                and getattr(model, 'rotation_output_format', None) == 'rotation_6d')

    def _spec_kwargs(self, transform_callable, targets) -> dict:
        model = self.movements_model
        out_idx, in_idx = get_common_indices(model.input_nodes, model.output_nodes)
        hips = model.input_nodes.get_hips_point()
        if isinstance(hips, (list, tuple)):
            hips_lane = -1                      # base_pose_loss.py:33-34
        elif isinstance(in_idx, slice):
            hips_lane = hips.value
        else:
            hips_lane = out_idx[list(in_idx).index(hips.value)]
        n2 = targets[self._gt2d_key(targets)].shape[-2] if self._gt2d_key(targets) else 26
        n3 = targets['absolute_pose_loc'].shape[-2] if 'absolute_pose_loc' in targets else 26
        kw = dict(mask_missing_joints=bool(self.mask_missing_joints), hips_lane=hips_lane,
                  gmap2d=joint_maps(out_idx, in_idx, n2), gmap3d=joint_maps(out_idx, in_idx, n3),
                  eval_slice=(model.eval_slice.start, model.eval_slice.stop))
        if transform_callable is None:
            kw['transform'] = 'none'
        else:
            hips_idx, neck_idx = transform_callable.extractor.points()
            kw.update(transform=transform_callable.kind, hips_idx=hips_idx, neck_idx=neck_idx,
                      near_zero=transform_callable.extractor.near_zero)
        return kw

    @staticmethod
    def _gt2d_key(targets):
        for k in ('projection_2d_transformed', 'projection_2d'):     # loc_2d.py:75-79
            if k in targets:
                return k
        return None

    # ---- two-launch train step (csrc/p2c_train.hip) -----------------------------------------------------------------
    def _fused_train_plan(self, frames, targets):
        """(PoseHeadSpec, gt2d, gt3d) when this batch can take ``ops.fused_train_step`` -- LinearAE with the 6-D rotation
        output on CARLA nodes, built-in transform, fusable losses, lean outputs, one clip per 16-sample tile, a small
        batch where only the one-workgroup-per-clip form applies (world motion or foreign target layouts: up to 2048 clips, the measured crossover with the separate kernels; ``P2C_FUSED_TRAIN_MAX_B`` sets a hard limit) -- else None.
        ``P2C_FUSED_TRAIN=0`` turns the path off. Everything but the two target tensors is a function of the configuration
        and the batch shape: it is worked out once per (shape, configuration) and cached."""
        import os
        model = self.movements_model
        if type(model) is not LinearAE or not frames.is_cuda or (self._datamodule is None and getattr(getattr(self, 'trainer', None), 'datamodule', None) is None):
            return None
        # the trainer stages every batch into the SAME tensor objects: the plan of the last call is still the plan
        fast = getattr(self, '_fused_plan_fast', None)
        if fast is not None and fast[0] is frames and fast[1] is targets and fast[2] == model.training:
            static = fast[3]
            if static is None:
                return None
            spec, gt2d_key, want3d = static
            return spec, (targets[gt2d_key] if gt2d_key else None), (targets.get('absolute_pose_loc') if want3d else None)
        transform_callable = self.datamodule.transform_callable
        key = (tuple(frames.shape), frames.device, frames.dtype, tuple(targets.keys()), os.environ.get('P2C_FUSED_TRAIN', '1'),
               os.environ.get('P2C_FUSED_TRAIN_MAX_B', ''), self.lean_train_outputs, type(model), model.eval_slice.start,
               model.eval_slice.stop, getattr(model, 'rotation_output_format', None), id(transform_callable),
               id(self.trajectory_model), bool(self.mask_missing_joints), model.fused_mlp, model.training, model.mlp_precision,
               tuple(targets[k].shape for k in ('projection_2d_transformed', 'projection_2d', 'absolute_pose_loc') if k in targets))
        cached = getattr(self, '_fused_plan_cache', None)
        if cached is None or cached[0] != key:
            cached = self._fused_plan_cache = (key, self._fused_train_plan_uncached(frames, targets, transform_callable))
        static = cached[1]
        self._fused_plan_fast = (frames, targets, model.training, static)
        if static is None:
            return None
        spec, gt2d_key, want3d = static
        return spec, (targets[gt2d_key] if gt2d_key else None), (targets.get('absolute_pose_loc') if want3d else None)

    def _fused_train_plan_uncached(self, frames, targets, transform_callable):
        import os
        from pedestrians_video_2_carla_amd import ops
        model = self.movements_model
        if os.environ.get('P2C_FUSED_TRAIN', '1') == '0' or type(model) is not LinearAE or not self.lean_train_outputs:
            return None
        if getattr(model, 'mlp_precision', 'fp32') != 'fp32':        # the two-launch step is exact fp32 only
            return None
        if not (frames.is_cuda and frames.dtype == torch.float32 and frames.ndim == 4):
            return None
        B, T = frames.shape[0], frames.shape[1]
        if model.input_nodes is not CARLA_SKELETON:
            return None
        max_b = os.environ.get('P2C_FUSED_TRAIN_MAX_B')
        if max_b is not None and B > int(max_b):
            return None
        if model.output_type not in (MovementsModelOutputType.pose_changes, MovementsModelOutputType.relative_rot) \
                or getattr(model, 'rotation_output_format', None) != 'rotation_6d':
            return None
        fa = model.fused_args(frames.device)
        if fa is None or not ops.train_step_supported(fa['dims'], T):
            return None
        if not self._fusable(transform_callable):
            return None
        names = {name for (name, *_r) in self._losses_to_calculate}
        if names - _FUSABLE_LOSSES:               # the two-launch step carries the location losses only
            return None
        gt2d_key = self._gt2d_key(targets) if 'loc_2d' in names else None
        kind = 'pose_changes_6d' if model.output_type == MovementsModelOutputType.pose_changes else 'relative_rot_6d'
        absolute = (self.trajectory_model.output_type == TrajectoryModelOutputType.loc_rot
                    and not bool(getattr(self.trajectory_model, 'is_identity', False)))
        spec = ops.PoseHeadSpec(kind=kind, world_absolute=absolute, **self._spec_kwargs(transform_callable, targets))
        if max_b is None and B > 2048:
            # Beyond ~2 000 clips the fused step pays only in its throughput form (csrc/p2c_train_stream.hip: a pair of wavefronts
            # per clip; 8192 clips: 345 us against 383 us for the separate kernels and 520 us for the one-workgroup-per-clip form),
            # which needs the plain case: no world motion, targets in the model's own joint layout.
            ident = list(range(len(spec.gmap2d)))
            plain = (bool(getattr(self.trajectory_model, 'is_identity', False)) and list(spec.gmap2d) == ident
                     and list(spec.gmap3d) == ident)
            if not plain:
                return None
        return spec, gt2d_key, ('loc_3d' in names)

    def _fused_train_step(self, frames, targets, stage):
        """The whole train step as one autograd node, or None when the separate kernels have to run."""
        from pedestrians_video_2_carla_amd import ops
        counts = getattr(self, '_pair_counts', None)
        if (stage != 'train' or counts is None or not torch.is_grad_enabled() or ops._DEFER_LOSS_FINALIZE != 2
                or counts.shape[0] != frames.shape[0]):
            return None
        plan = self._fused_train_plan(frames, targets)
        if plan is None:
            return None
        spec, gt2d, gt3d = plan
        model, traj = self.movements_model, self.trajectory_model
        self.projection._check_ready(frames)
        if bool(getattr(traj, 'is_identity', False)):
            dloc = drot = None
        else:
            dloc, drot = traj(frames, targets if self.training and traj.needs_targets else None)
        fa = model.fused_args(frames.device)
        losses = ops.fused_train_step(frames, fa['weights'], fa['biases'], spec, self.projection._skel_type, counts,
                                      dloc=dloc, drot=drot, gt2d=gt2d, gt3d=gt3d, sinks=fa['sinks'], image=fa['image'],
                                      image_is_current=fa['image_is_current'], fused_optimizer=fa['fused_optimizer'])
        eval_slice = (slice(None), model.eval_slice)
        sliced = {'_fused': FusedLosses(losses, model.input_nodes, model.output_nodes, bool(self.mask_missing_joints),
                                        gt2d is not None, gt3d is not None),
                  'projection_2d': None, 'pose_inputs': None,
                  'world_loc_inputs': dloc[eval_slice] if dloc is not None else None,
                  'world_rot_inputs': drot[eval_slice] if drot is not None else None,
                  'inputs': frames[eval_slice], 'targets': {k: v[eval_slice] for k, v in targets.items()}}
        if self.datamodule.transform_callable is not None:
            sliced['projection_2d_transformed'] = None
        for k in set(list(_PROJECTION_KEYS) + self._crucial_keys):
            sliced.setdefault(k, None)
        if 'world_loc_changes' in targets and 'world_rot_changes' in targets:      # pose_lifting.py:186-194
            target_world_loc, target_world_rot = calculate_world_from_changes(
                frames.shape, frames.device, targets['world_loc_changes'], targets['world_rot_changes'])
            sliced['targets']['world_loc'] = target_world_loc[eval_slice]
            sliced['targets']['world_rot'] = target_world_rot[eval_slice]
        return sliced

    # ---- the step ---------------------------------------------------------------------------------------------------
    def _inner_step(self, frames, targets, edge_index=None, batch_vector=None, stage='train'):
        fused = self._fused_train_step(frames, targets, stage)
        if fused is not None:
            return fused
        model, traj = self.movements_model, self.trajectory_model
        pose_inputs = model(frames, targets if self.training and model.needs_targets else None,
                            edge_index=None, batch_vector=None)
        lean = self.lean_train_outputs and stage == 'train'
        identity_world = bool(getattr(traj, 'is_identity', False))
        if identity_world and lean:
            world_loc_inputs = world_rot_inputs = None           # ZeroTrajectory constant-folded
        else:
            world_loc_inputs, world_rot_inputs = traj(frames, targets if self.training and traj.needs_targets else None)

        transform_callable = self.datamodule.transform_callable
        eval_slice = (slice(None), model.eval_slice)
        sliced = {}

        if self._fusable(transform_callable):
            names = {name for (name, *_r) in self._losses_to_calculate}
            gt2d_key = self._gt2d_key(targets) if 'loc_2d' in names else None
            gt2d = targets[gt2d_key] if gt2d_key else None
            gt3d = targets.get('absolute_pose_loc') if 'loc_3d' in names else None
            gt_rot = targets.get('absolute_pose_rot') if 'rot_3d' in names else None
            spec_kwargs = self._spec_kwargs(transform_callable, targets)
            want = ()
            if not lean:
                from pedestrians_video_2_carla_amd import ops
                probe = ops.PoseHeadSpec(kind=self.projection.kernel_kind(pose_inputs), **spec_kwargs)
                want = ops.available_outputs(probe, not identity_world)
            y = pose_inputs[0] if isinstance(pose_inputs, tuple) else pose_inputs
            losses, outs = self.projection.fused_losses(pose_inputs, world_loc_inputs, world_rot_inputs,
                                                        identity_world, spec_kwargs, gt2d, gt3d, want, gt_rot)
            sliced['_fused'] = FusedLosses(losses, model.input_nodes, model.output_nodes,
                                           bool(self.mask_missing_joints), gt2d is not None, gt3d is not None,
                                           getattr(losses, 'rot_3d', None))
            if not lean and identity_world:
                outs['world_loc'], outs['world_rot'] = world_loc_inputs, world_rot_inputs   # zeros / identity
            if not lean and isinstance(pose_inputs, tuple):
                outs['absolute_pose_rot'] = pose_inputs[1]
            if 'pose_changes' in outs:
                pose_inputs = outs['pose_changes']          # the reference's API tensor (B,T,J,3,3)
            elif lean and y.ndim == 4 and y.shape[-1] == 6:
                pose_inputs = None                           # 6-D network output; matrices not materialised
            sliced['projection_2d'] = outs['projection_2d'][eval_slice] if 'projection_2d' in outs else None
            if transform_callable is not None:
                sliced['projection_2d_transformed'] = (outs['projection_2d_transformed'][eval_slice]
                                                       if 'projection_2d_transformed' in outs else None)
            projection_outputs_dict = {k: outs.get(k) for k in _PROJECTION_KEYS}
        else:
            projection_2d, projection_outputs_dict = self.projection(pose_inputs, world_loc_inputs, world_rot_inputs,
                                                                     identity_world=identity_world)
            sliced['projection_2d'] = projection_2d[eval_slice]
            if transform_callable is not None:
                sliced['projection_2d_transformed'] = transform_callable(projection_2d[eval_slice])

        sliced['pose_inputs'] = (tuple(v[eval_slice] for v in pose_inputs) if isinstance(pose_inputs, tuple)
                                 else (pose_inputs[eval_slice] if pose_inputs is not None else None))
        sliced['world_loc_inputs'] = world_loc_inputs[eval_slice] if world_loc_inputs is not None else None
        sliced['world_rot_inputs'] = world_rot_inputs[eval_slice] if world_rot_inputs is not None else None
        sliced['inputs'] = frames[eval_slice]
        sliced['targets'] = {k: v[eval_slice] for k, v in targets.items()}
        for k in set(list(projection_outputs_dict.keys()) + self._crucial_keys):
            if k not in sliced:
                v = projection_outputs_dict.get(k)
                sliced[k] = v[eval_slice] if v is not None else None

        if 'world_loc_changes' in targets and 'world_rot_changes' in targets:      # pose_lifting.py:186-194
            target_world_loc, target_world_rot = calculate_world_from_changes(
                frames.shape, frames.device, targets['world_loc_changes'], targets['world_rot_changes'])
            sliced['targets']['world_loc'] = target_world_loc[eval_slice]
            sliced['targets']['world_rot'] = target_world_rot[eval_slice]
        return sliced
