#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own hot-path modules (build container only).

The reference (/root/reference, Python) imports here once its absent *third-party* packages are given in-memory
stand-ins: pedestrians_scenarios (Skeleton/CARLA_SKELETON enums, deepcopy helpers), pytorch3d 0.6.0
(rotation conversions, look_at_view_transform, screen-space PerspectiveCameras), cameratransform, pytorch_lightning's
rank_zero_warn. The stand-ins re-state those packages' public definitions -- they are this build's code, not the
reference's. Every module of the reference itself (ProjectionModule, P3dPose, P3dPoseProjection, ControlledPedestrian,
Normalizer + extractors, ReferenceSkeletonsDeNormalizer, LossModes / Loc2DPoseLoss / loc_3d / loc_2d_3d,
calculate_world_from_changes, get_common_indices, LinearAE, Seq2SeqEmbeddings) is imported and executed unmodified.

Outputs are data only (inputs + expected outputs), small (B=4, T=16). /root/reference never travels to the GPU box;
tests read the committed .npz files.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_SRC = '/root/reference/src'

warnings.filterwarnings('ignore')


# ------------------------------------------------------------------------------------------------------------
# stand-ins for third-party packages (NOT for any reference module)
# ------------------------------------------------------------------------------------------------------------
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent:
        if parent not in sys.modules:
            _module(parent)
        setattr(sys.modules[parent], child, m)
    return m


def install_standins():
    from pedestrians_video_2_carla_amd.data.base.skeleton import Skeleton
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from oracle import pose_head as O

    # --- pedestrians_scenarios 0.0.1 -------------------------------------------------------------------------
    _module('pedestrians_scenarios')
    _module('pedestrians_scenarios.karma')
    _module('pedestrians_scenarios.karma.pose')
    _module('pedestrians_scenarios.karma.pose.skeleton', Skeleton=Skeleton, CARLA_SKELETON=CARLA_SKELETON)
    _module('pedestrians_scenarios.karma.pose.types', PoseDict=dict)
    _module('pedestrians_scenarios.karma.utils')

    def deepcopy_location(l):
        return type(l)(x=l.x, y=l.y, z=l.z)

    def deepcopy_rotation(r):
        return type(r)(pitch=r.pitch, yaw=r.yaw, roll=r.roll)

    def deepcopy_transform(t):
        return type(t)(location=deepcopy_location(t.location), rotation=deepcopy_rotation(t.rotation))

    _module('pedestrians_scenarios.karma.utils.deepcopy', deepcopy_location=deepcopy_location,
            deepcopy_rotation=deepcopy_rotation, deepcopy_transform=deepcopy_transform)
    _module('pedestrians_scenarios.karma.utils.rotations', mul_carla_rotations=None)

    # --- pytorch3d 0.6.0 --------------------------------------------------------------------------------------
    def euler_angles_to_matrix(euler_angles, convention):
        assert convention == 'XYZ'
        return O.euler_angles_to_matrix_xyz(euler_angles)

    def matrix_to_euler_angles(*a, **k):
        raise NotImplementedError('not on the hot path')

    def look_at_view_transform(eye, at, up):
        eye, at, up = (torch.tensor(v, dtype=torch.float32) for v in (eye, at, up))
        nrm = torch.nn.functional.normalize
        z = nrm(at - eye, dim=-1)
        x = nrm(torch.cross(up, z, dim=-1), dim=-1)
        y = nrm(torch.cross(z, x, dim=-1), dim=-1)
        R = torch.stack((x, y, z), dim=-1)                  # columns x, y, z
        T = -(R.transpose(1, 2) @ eye[..., None])[..., 0]
        return R, T

    class PerspectiveCameras:
        def __init__(self, device, in_ndc, focal_length, principal_point, image_size, R, T):
            assert in_ndc is False
            self.device = device
            self.f = float(np.asarray(focal_length).reshape(-1)[0])
            self.pp = torch.tensor(principal_point, dtype=torch.float32, device=device).reshape(-1, 2)
            self.R = R.to(device)
            self.T = T.to(device)

        def transform_points_screen(self, points):
            view = points @ self.R + self.T[:, None, :]
            X, Y, Z = view[..., 0], view[..., 1], view[..., 2]
            return torch.stack((self.pp[:, 0:1] - self.f * X / Z, self.pp[:, 1:2] - self.f * Y / Z, 1.0 / Z), -1)

    _module('pytorch3d')
    _module('pytorch3d.transforms', euler_angles_to_matrix=euler_angles_to_matrix,
            rotation_6d_to_matrix=O.rotation_6d_to_matrix, matrix_to_rotation_6d=O.matrix_to_rotation_6d)
    _module('pytorch3d.transforms.rotation_conversions', euler_angles_to_matrix=euler_angles_to_matrix,
            matrix_to_euler_angles=matrix_to_euler_angles, rotation_6d_to_matrix=O.rotation_6d_to_matrix,
            matrix_to_rotation_6d=O.matrix_to_rotation_6d)
    _module('pytorch3d.transforms.transform3d', Rotate=None, Translate=None)
    _module('pytorch3d.renderer')
    _module('pytorch3d.renderer.cameras', PerspectiveCameras=PerspectiveCameras,
            look_at_view_transform=look_at_view_transform)

    # --- misc ---------------------------------------------------------------------------------------------------
    _module('cameratransform')
    _module('pytorch_lightning')
    _module('pytorch_lightning.utilities')
    _module('pytorch_lightning.utilities.warnings', rank_zero_warn=lambda *a, **k: None)


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, len(out), 'arrays')


def main():
    if not os.path.isdir(REF_SRC):
        sys.exit('reference tree not present: the committed .npz files are the artefact to use')
    install_standins()
    sys.path.insert(0, REF_SRC)
    torch.manual_seed(22742)

    import pedestrians_video_2_carla as pkg  # noqa: F401
    from pedestrians_video_2_carla.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla.data.openpose.skeleton import BODY_25_SKELETON, COCO_SKELETON
    from pedestrians_video_2_carla.data.base.skeleton import get_common_indices
    from pedestrians_video_2_carla.data.carla import reference as ref_tables
    from pedestrians_video_2_carla.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla.modules.layers.projection import ProjectionModule
    from pedestrians_video_2_carla.transforms.pose.normalization import Normalizer
    from pedestrians_video_2_carla.transforms.pose.normalization.hips_neck_extractor import HipsNeckExtractor
    from pedestrians_video_2_carla.transforms.pose.normalization.bbox_extractor import BBoxExtractor
    from pedestrians_video_2_carla.transforms.pose.normalization.hips_neck_bbox_fallback_extractor import \
        HipsNeckBBoxFallbackExtractor
    from pedestrians_video_2_carla.transforms.pose.normalization.reference_skeletons_denormalizer import \
        ReferenceSkeletonsDeNormalizer
    from pedestrians_video_2_carla.loss import LossModes
    from pedestrians_video_2_carla.utils.world import calculate_world_from_changes
    from pedestrians_video_2_carla.utils.tensors import get_bboxes
    from pytorch3d.transforms import rotation_6d_to_matrix, euler_angles_to_matrix

    types_ = ref_tables.CARLA_REFERENCE_SKELETON_TYPES
    B, T, J = 4, 16, 26
    meta = {'age': [a for a, _ in types_], 'gender': [g for _, g in types_]}

    # ---- 0. reference tables ------------------------------------------------------------------------------------
    rel_loc, rel_rot = ref_tables.get_relative_tensors()
    abs_loc, abs_rot = ref_tables.get_absolute_tensors()
    projections = ref_tables.get_projections()
    npz('reference_tables', rel_loc=rel_loc, rel_rot=rel_rot, abs_loc=abs_loc, abs_rot=abs_rot,
        projections=projections)

    def run_losses(sliced, targets, in_nodes=CARLA_SKELETON, out_nodes=CARLA_SKELETON, mask=True):
        res = {}
        l2d = LossModes.loc_2d.value[0](criterion=LossModes.loc_2d.value[1], input_nodes=in_nodes,
                                        output_nodes=out_nodes, mask_missing_joints=mask)
        res['loc_2d'] = l2d(targets=targets, **sliced)
        res['loc_3d'] = LossModes.loc_3d.value[0](criterion=LossModes.loc_3d.value[1], input_nodes=in_nodes,
                                                  output_nodes=out_nodes, targets=targets, **sliced)
        res['loc_2d_3d'] = LossModes.loc_2d_3d.value[0](requirements=res)
        return res

    def make_targets(seed, missing=0.0):
        g = torch.Generator().manual_seed(seed)
        ang = (torch.rand(B, T, J, 3, generator=g) * 2 - 1) * np.deg2rad(8.0)
        pm = ProjectionModule(movements_output_type=MT.pose_changes)
        pm.on_batch_start((torch.zeros(B, T, J, 2), None, meta), 0)
        proj, d = pm(euler_angles_to_matrix(ang, 'XYZ'), None, None)
        norm = Normalizer(HipsNeckBBoxFallbackExtractor(CARLA_SKELETON))
        p2t = norm(proj[..., :2].clone())
        if missing > 0:
            miss = torch.rand(B, T, J, generator=g) < missing
            p2t[miss] = 0.0
        return {'projection_2d': proj[..., :2].clone(), 'projection_2d_transformed': p2t,
                'absolute_pose_loc': d['absolute_pose_loc'].clone()}

    # ---- 1. pose_changes variant (6D model output), with and without missing gt joints, with world motion ------
    for tag, missing, world in (('pose_changes', 0.0, False), ('pose_changes_missing', 0.15, False),
                                ('pose_changes_world', 0.0, True)):
        g = torch.Generator().manual_seed(1 if not world else 2)
        y6d = torch.randn(B, T, J, 6, generator=g)
        y6d[..., 0] += 2.0
        y6d[..., 4] += 2.0            # near-identity-ish but far from degenerate
        y6d.requires_grad_(True)
        targets = make_targets(11, missing)
        dloc = drot = None
        if world:
            dloc = torch.randn(B, T, 3, generator=g) * 0.05
            drot = euler_angles_to_matrix((torch.rand(B, T, 3, generator=g) * 2 - 1) * 0.1, 'XYZ')
        pm = ProjectionModule(movements_output_type=MT.pose_changes)
        pm.on_batch_start((torch.zeros(B, T, J, 2), None, meta), 0)
        changes = rotation_6d_to_matrix(y6d)
        proj, d = pm(changes, dloc, drot)
        norm = Normalizer(HipsNeckBBoxFallbackExtractor(CARLA_SKELETON))
        proj_t = norm(proj)
        sliced = {'projection_2d': proj, 'projection_2d_transformed': proj_t, **d}
        res = run_losses(sliced, targets)
        res['loc_2d_3d'].backward()
        npz(tag, y6d=y6d, pose_changes=changes, skel_type=np.arange(4),
            dloc=dloc if world else np.zeros(0), drot=drot if world else np.zeros(0),
            projection_2d=proj, projection_2d_transformed=proj_t, shift=norm.shift, scale=norm.scale,
            relative_pose_loc=d['relative_pose_loc'], relative_pose_rot=d['relative_pose_rot'],
            absolute_pose_loc=d['absolute_pose_loc'], absolute_pose_rot=d['absolute_pose_rot'],
            world_loc=d['world_loc'], world_rot=d['world_rot'],
            gt_projection_2d_transformed=targets['projection_2d_transformed'],
            gt_absolute_pose_loc=targets['absolute_pose_loc'],
            loc_2d=res['loc_2d'], loc_3d=res['loc_3d'], loc_2d_3d=res['loc_2d_3d'], grad_y=y6d.grad)

    # ---- 2. absolute_loc variant -----------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(3)
    targets = make_targets(12)
    y = (targets['absolute_pose_loc'] * 0.7 + 0.1 * torch.randn(B, T, J, 3, generator=g) + 0.3).requires_grad_(True)
    pm = ProjectionModule(movements_output_type=MT.absolute_loc)
    pm.on_batch_start((torch.zeros(B, T, J, 2), None, meta), 0)
    proj, d = pm(y, None, None)
    norm = Normalizer(HipsNeckBBoxFallbackExtractor(CARLA_SKELETON))
    proj_t = norm(proj)
    res = run_losses({'projection_2d': proj, 'projection_2d_transformed': proj_t, **d}, targets)
    res['loc_2d_3d'].backward()
    npz('absolute_loc', y=y, skel_type=np.arange(4), projection_2d=proj, projection_2d_transformed=proj_t,
        absolute_pose_loc=d['absolute_pose_loc'], world_loc=d['world_loc'], world_rot=d['world_rot'],
        gt_projection_2d_transformed=targets['projection_2d_transformed'],
        gt_absolute_pose_loc=targets['absolute_pose_loc'],
        loc_2d=res['loc_2d'], loc_3d=res['loc_3d'], loc_2d_3d=res['loc_2d_3d'], grad_y=y.grad)

    # ---- 3. relative_rot variant -----------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(4)
    y6d = torch.randn(B, T, J, 6, generator=g)
    pm = ProjectionModule(movements_output_type=MT.relative_rot)
    pm.on_batch_start((torch.zeros(B, T, J, 2), None, meta), 0)
    proj, d = pm(rotation_6d_to_matrix(y6d), None, None)
    npz('relative_rot', y6d=y6d, skel_type=np.arange(4), projection_2d=proj,
        absolute_pose_loc=d['absolute_pose_loc'], absolute_pose_rot=d['absolute_pose_rot'],
        relative_pose_loc=d['relative_pose_loc'])

    # ---- 4. normalisers incl. fallback paths --------------------------------------------------------------------------
    g = torch.Generator().manual_seed(5)
    base = make_targets(13)['projection_2d']                       # pixels
    cases = base.clone()                                           # (4,16,26,2)
    cases[0, 0:4, 1] = 0.0                                         # hips missing
    cases[0, 4:8, 8] = 0.0                                         # neck missing
    cases[1, 0:4, [1, 8]] = 0.0                                    # both
    cases[1, 4:8] = 0.0                                            # all joints missing
    cases[2, 0:4, 3:20] = 0.0                                      # many missing, hips present
    cases[2, 4:8, 1] = -5.0                                        # negative hips == "missing" for < near_zero test
    cases[3, 0:4] = cases[3, 0:4] - 450.0                          # partially negative coordinates
    conf = torch.rand(B, T, J, 1, generator=g)
    conf[3, 8:12, 5:9] = 0.0                                       # zero-confidence joints -> xy zeroed
    cases3 = torch.cat((cases, conf), -1)
    outs = {}
    for name, ext in (('hips_neck', HipsNeckExtractor), ('bbox', BBoxExtractor),
                      ('hips_neck_bbox', HipsNeckBBoxFallbackExtractor)):
        n = Normalizer(ext(CARLA_SKELETON))
        outs[f'{name}_out2'] = n(cases.clone())
        outs[f'{name}_shift2'] = n.shift
        outs[f'{name}_scale2'] = n.scale
        outs[f'{name}_out3'] = n(cases3.clone())
    outs['bboxes'] = get_bboxes(cases)
    npz('normalizers', cases=cases, cases3=cases3, **outs)

    # normaliser gradients through the fallback path (bbox scale) and the regular path
    x = cases.clone()
    x[1, 4:8] = base[1, 4:8]                                        # drop the all-missing rows (nan grads in reference)
    x.requires_grad_(True)
    n = Normalizer(HipsNeckBBoxFallbackExtractor(CARLA_SKELETON))
    w = torch.randn(B, T, J, 2, generator=g)
    (n(x) * w).sum().backward()
    npz('normalizer_grad', x=x, w=w, grad=x.grad)

    # ---- 5. reference-skeleton de-normaliser ---------------------------------------------------------------------------
    dn = ReferenceSkeletonsDeNormalizer()
    xin = abs_loc[:, None].repeat(1, 3, 1, 1) * torch.tensor([0.5, 1.0, 2.5]).reshape(1, 3, 1, 1) + 0.25
    npz('denormalizer', x=xin, out=dn.from_abs(xin, meta, autonormalize=True))

    # ---- 6. joint index maps ---------------------------------------------------------------------------------------------
    def as_arr(v):
        return np.arange(26) if isinstance(v, slice) else np.asarray(v)
    pairs = {}
    for a_name, a in (('body25', BODY_25_SKELETON), ('coco', COCO_SKELETON), ('carla', CARLA_SKELETON)):
        for b_name, b in (('body25', BODY_25_SKELETON), ('coco', COCO_SKELETON), ('carla', CARLA_SKELETON)):
            if a is b:
                continue
            o, i = get_common_indices(input_nodes=a, output_nodes=b)
            pairs[f'in_{a_name}__out_{b_name}__out_idx'] = as_arr(o)
            pairs[f'in_{a_name}__out_{b_name}__in_idx'] = as_arr(i)
    npz('common_indices', **pairs)

    # ---- 7. world transform with real rotations -----------------------------------------------------------------------------
    g = torch.Generator().manual_seed(6)
    dloc = torch.randn(3, 10, 3, generator=g)
    drot = euler_angles_to_matrix(torch.rand(3, 10, 3, generator=g), 'XYZ')
    il = torch.randn(3, 3, generator=g)
    ir = euler_angles_to_matrix(torch.rand(3, 3, generator=g), 'XYZ')
    wl, wr = calculate_world_from_changes((3, 10, 26, 3), torch.device('cpu'), dloc, drot, il, ir)
    npz('world', dloc=dloc, drot=drot, init_loc=il, init_rot=ir, world_loc=wl, world_rot=wr)

    # ---- 8. model plugins under a fixed state_dict ------------------------------------------------------------------------------
    from pedestrians_video_2_carla.modules.movements.linear_ae.linear_ae import LinearAE
    from pedestrians_video_2_carla.modules.movements.seq2seq.seq2seq_embeddings import Seq2SeqEmbeddings
    frames = make_targets(14)['projection_2d_transformed']
    for name, cls, kw in (
            ('linear_ae_pose_changes', LinearAE, dict(movements_output_type=MT.pose_changes)),
            ('linear_ae_absolute_loc', LinearAE, dict(movements_output_type=MT.absolute_loc)),
            ('linear_ae_pose_2d', LinearAE, dict(movements_output_type=MT.pose_2d)),
            ('seq2seq_embeddings_pose_2d', Seq2SeqEmbeddings, dict(movements_output_type=MT.pose_2d)),
    ):
        torch.manual_seed(22742)
        model = cls(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, **kw).eval()
        with torch.no_grad():
            out = model(frames)
        sd = {('sd__' + k): v for k, v in model.state_dict().items()}
        npz('model_' + name, frames=frames, out=out, n_params=sum(p.numel() for p in model.parameters()), **sd)


def golden_metrics():
    """Section 9: the reference's own MPJPE / MRPE / PCK classes (metrics/mpjpe.py, mrpe.py, pck.py), run on
    two batches each. torchmetrics is absent from the image: its ``Metric`` base gets a stand-in that keeps ``add_state``
    values as attributes (update / compute of the reference run unmodified)."""
    class Metric(torch.nn.Module):
        def __init__(self, dist_sync_on_step=False, **kwargs):
            super().__init__()

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, default.clone())

    _module('torchmetrics', Metric=Metric)
    from pedestrians_video_2_carla.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla.data.openpose.skeleton import BODY_25_SKELETON
    from pedestrians_video_2_carla.metrics.mpjpe import MPJPE
    from pedestrians_video_2_carla.metrics.mrpe import MRPE
    from pedestrians_video_2_carla.metrics.pck import PCK
    g = torch.Generator().manual_seed(99)
    out = {}
    batches = []
    for k in range(2):
        B = 3 + k
        batches.append(dict(
            pred_abs=torch.randn(B, 16, 26, 3, generator=g), gt_abs=torch.randn(B, 16, 26, 3, generator=g),
            gt_abs_b25=torch.randn(B, 16, 25, 3, generator=g),
            pred_wlc=torch.randn(B, 16, 3, generator=g) * 0.05, gt_wlc=torch.randn(B, 16, 3, generator=g) * 0.05,
            pred_p2d=torch.rand(B, 16, 26, 2, generator=g) * 400 + 100, gt_p2d=torch.rand(B, 16, 26, 2, generator=g) * 400 + 100))
        batches[-1]['pred_p2d'] = batches[-1]['gt_p2d'] + torch.randn(B, 16, 26, 2, generator=g) * 12
        batches[-1]['gt_p2d'][0, 0, 5] = 0          # a missing ground-truth joint (masked out of PCK)
        for name, v in batches[-1].items():
            out[f'b{k}_{name}'] = v
    m1, m2 = MPJPE(), MPJPE(input_nodes=BODY_25_SKELETON, output_nodes=CARLA_SKELETON)
    m3 = MRPE()
    pck_bbox, pck_hn = PCK(), PCK(get_normalization_tensor='hn', threshold=0.2)
    for b in batches:
        m1.update({'absolute_pose_loc': b['pred_abs']}, {'absolute_pose_loc': b['gt_abs']})
        m2.update({'absolute_pose_loc': b['pred_abs']}, {'absolute_pose_loc': b['gt_abs_b25']})
        m3.update({'absolute_pose_loc': b['pred_abs'], 'world_loc_changes': b['pred_wlc']},
                  {'absolute_pose_loc': b['gt_abs'], 'world_loc_changes': b['gt_wlc']})
        for p in (pck_bbox, pck_hn):
            p.update({'projection_2d': b['pred_p2d']}, {'projection_2d': b['gt_p2d']})
    npz('metrics', mpjpe=m1.compute(), mpjpe_body25=m2.compute(), mrpe=m3.compute(), pck_bbox=pck_bbox.compute(),
        pck_hn=pck_hn.compute(), pck_bbox_correct=pck_bbox.correct, pck_bbox_total=pck_bbox.total, **out)


def golden_collate():
    """collate.npz: the reference's dataset-side input pipeline run clip by clip (BaseDataset.__getitem__:
    process_projection_2d -> process_confidence -> _map_nodes, data/base/base_dataset.py:206-234) on small in-memory
    sets. The random draws are recorded by replaying the dataset generator in the order the reference consumes it
    (flip, rotation, noise, missing) so that the oracle and the HIP kernel can be fed the same numbers."""
    if not os.path.isdir(REF_SRC):
        sys.exit('reference tree not present: the committed .npz files are the artefact to use')
    install_standins()
    _module('h5py')                      # third-party, only touched by _load_data (overridden below)
    try:
        import importlib_metadata  # noqa: F401
    except ImportError:
        _module('importlib_metadata', metadata=None)
    sys.path.insert(0, REF_SRC)
    from pedestrians_video_2_carla.data.base.base_dataset import BaseDataset
    from pedestrians_video_2_carla.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla.data.openpose.skeleton import BODY_25_SKELETON, COCO_SKELETON
    from pedestrians_video_2_carla.transforms.pose.normalization import Normalizer
    from pedestrians_video_2_carla.transforms.pose.normalization.hips_neck_extractor import HipsNeckExtractor
    from pedestrians_video_2_carla.transforms.pose.normalization.bbox_extractor import BBoxExtractor
    from pedestrians_video_2_carla.transforms.pose.normalization.hips_neck_bbox_fallback_extractor import \
        HipsNeckBBoxFallbackExtractor

    class MemoryDataset(BaseDataset):
        def __init__(self, clips, bboxes=None, clip_size=None, **kwargs):
            self._clips, self._bboxes, self._clip_size = clips, bboxes, clip_size
            super().__init__(set_filepath=None, skip_metadata=True, **kwargs)

        def _load_data(self, ignore_metadata=False):
            self.projection_2d = self._clips
            self.meta = [{}] * len(self._clips)

        def _get_targets(self, idx, raw_projection_2d, intermediate_outputs):
            return {} if self._bboxes is None else {'bboxes': torch.from_numpy(self._bboxes[idx])}

        def _get_meta(self, idx):
            if self._clip_size is None:
                return {'clip_width': float('nan'), 'clip_height': float('nan')}
            return {'clip_width': float(self._clip_size[idx, 0]), 'clip_height': float(self._clip_size[idx, 1])}

    def synth(nodes, N, T, C, seed, missing=0.1):
        g = torch.Generator().manual_seed(seed)
        J = len(nodes)
        centre = torch.rand(N, 1, 1, 2, generator=g) * torch.tensor([900., 500.]) + torch.tensor([300., 200.])
        walk = torch.cumsum(torch.randn(N, T, 1, 2, generator=g) * 3.0, dim=1)
        pts = centre + walk + torch.randn(N, 1, J, 2, generator=g) * torch.tensor([40., 90.]) \
            + torch.randn(N, T, J, 2, generator=g) * 2.0
        gone = torch.rand(N, T, J, generator=g) < missing
        if C == 3:
            conf = torch.rand(N, T, J, 1, generator=g) * 0.9 + 0.05
            pts = torch.cat((pts, conf), dim=-1)
        pts[gone] = 0.0
        return pts.numpy().astype(np.float32)

    def boxes_of(clips, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.from_numpy(clips[..., :2]).clone()
        seen = ~torch.all(x < 1e-5, dim=-1, keepdim=True)
        lo = torch.where(seen, x, torch.full_like(x, float('inf'))).amin(dim=-2)
        hi = torch.where(seen, x, torch.full_like(x, float('-inf'))).amax(dim=-2)
        pad = torch.rand(*lo.shape, generator=g) * 10.0
        return torch.stack((lo - pad, hi + pad), dim=-2).numpy().astype(np.float32)      # (N,T,2,2)

    hn = lambda nodes: Normalizer(HipsNeckExtractor(nodes))
    hnb = lambda nodes: Normalizer(HipsNeckBBoxFallbackExtractor(nodes))
    bb = lambda nodes: Normalizer(BBoxExtractor(nodes))
    N, T = 3, 8
    cases = {
        # name: data nodes, input nodes, channels, has bboxes / clip size, dataset kwargs
        'body25_full': (BODY_25_SKELETON, CARLA_SKELETON, 3, True, True,
                        dict(transform=hnb(BODY_25_SKELETON), noise='gaussian', noise_param=2.5, augment_flip=1.0,
                             augment_rotate=15.0, missing_joint_probabilities=[0.1], is_training=True)),
        'body25_nosize': (BODY_25_SKELETON, CARLA_SKELETON, 3, True, False,
                          dict(transform=hn(BODY_25_SKELETON), augment_flip=1.0, is_training=True,
                               needs_confidence=True)),
        'carla_2ch': (CARLA_SKELETON, CARLA_SKELETON, 2, False, False,
                      dict(transform=hn(CARLA_SKELETON), noise='gaussian', noise_param=1.0, augment_flip=0.5,
                           augment_rotate=True, is_training=True,
                           missing_joint_probabilities=[0.02 * (j % 5) for j in range(26)])),
        'coco_eval': (COCO_SKELETON, CARLA_SKELETON, 3, False, False,
                      dict(transform=bb(COCO_SKELETON), missing_joint_probabilities=[0.2], needs_confidence=True,
                           augment_flip=True, augment_rotate=True, is_training=False)),
        'carla_plain': (CARLA_SKELETON, CARLA_SKELETON, 2, False, False, dict(transform=None, is_training=True)),
    }
    out = {}
    for ci, (name, (dn, inn, C, has_box, has_size, kw)) in enumerate(cases.items()):
        clips = synth(dn, N, T, C, seed=100 + ci)
        boxes = boxes_of(clips, seed=200 + ci) if has_box else None
        size = np.array([[1920., 1080.], [1280., 720.], [1920., 1080.]], np.float32) if has_size else None
        ds = MemoryDataset(clips, boxes, size, data_nodes=dn, input_nodes=inn, **kw)
        ds.generator.manual_seed(4242 + ci)
        twin = torch.Generator().manual_seed(4242 + ci)
        J = len(dn)
        rec = {k: [] for k in ('is_flipped', 'rotation', 'noise', 'miss_u')}
        frames, tg = [], {}
        for n in range(N):
            f, t, _ = ds[n]
            frames.append(f)
            for k, v in t.items():
                tg.setdefault(k, []).append(torch.as_tensor(v))
            # replay the draws of this clip on the twin generator, in the reference's order of consumption
            if ds.needs_augmentation:
                if ds.augmentation.flip is not None:
                    rec['is_flipped'].append(torch.rand((1,), generator=twin) < ds.augmentation.flip.prob)
                if ds.augmentation.rotate is not None:
                    rec['rotation'].append((torch.rand((1,), generator=twin) * 2 - 1)
                                           * ds.augmentation.rotate.max_rotation_angle)
            if ds.needs_noise:
                rec['noise'].append(torch.normal(mean=0.0, std=ds.noise_param, size=(T, J, 2), generator=twin))
            if ds.needs_missing_points:
                rec['miss_u'].append(torch.rand((T, J), generator=twin))
        assert torch.equal(twin.get_state(), ds.generator.get_state()), name     # the replay consumed exactly the same
        out[name + '/raw'] = clips
        if boxes is not None:
            out[name + '/bboxes_in'] = boxes
        if size is not None:
            out[name + '/clip_size'] = size
        out[name + '/frames'] = torch.stack(frames)
        for k, v in tg.items():
            out[name + '/t_' + k] = torch.stack([x.reshape(x.shape) for x in v])
        for k, v in rec.items():
            if v:
                out[name + '/' + k] = torch.stack(v) if k in ('noise', 'miss_u') else torch.cat(v)
        if ds.needs_missing_points:
            out[name + '/miss_prob'] = np.asarray(ds.missing_joint_probabilities, np.float32)
    npz('collate', **out)


def golden_losses_extra():
    """losses_extra.npz: the reference's cum_pose_changes function and PerJointLoc2DPoseLoss class on small inputs."""
    if not os.path.isdir(REF_SRC):
        sys.exit('reference tree not present: the committed .npz files are the artefact to use')
    install_standins()
    sys.path.insert(0, REF_SRC)
    from pedestrians_video_2_carla.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla.data.openpose.skeleton import BODY_25_SKELETON
    from pedestrians_video_2_carla.loss import LossModes
    from pytorch3d.transforms import euler_angles_to_matrix
    g = torch.Generator().manual_seed(77)
    B, T, J = 3, 16, 26
    pred = euler_angles_to_matrix((torch.rand(B, T, J, 3, generator=g) * 2 - 1) * 0.2, 'XYZ')
    tgt = euler_angles_to_matrix((torch.rand(B, T, J, 3, generator=g) * 2 - 1) * 0.2, 'XYZ')
    fn, crit = LossModes.cum_pose_changes.value
    out = {'cum_pred': pred, 'cum_tgt': tgt, 'cum_loss': fn(criterion=crit, pose_inputs=pred, targets={'pose_changes': tgt})}
    weights = (torch.rand(25, generator=g) + 0.5).tolist()
    for name, inn, outn, wts in (('carla', CARLA_SKELETON, CARLA_SKELETON, (torch.rand(26, generator=g) + 0.5).tolist()),
                                 ('b25', BODY_25_SKELETON, CARLA_SKELETON, weights)):
        p2 = torch.randn(B, T, len(outn), 3, generator=g)
        gt = torch.randn(B, T, len(inn), 2, generator=g)
        gt[torch.rand(B, T, len(inn), generator=g) < 0.15] = 0.0
        cls, crit = LossModes.per_joint_loc_2d.value
        for mask in (True, False):
            loss = cls(criterion=crit, input_nodes=inn, output_nodes=outn, mask_missing_joints=mask, loss_params=wts)
            out[f'pj_{name}_loss_mask{int(mask)}'] = loss(projection_2d_transformed=p2,
                                                         targets={'projection_2d_transformed': gt})
        out[f'pj_{name}_pred'], out[f'pj_{name}_gt'], out[f'pj_{name}_weights'] = p2, gt, torch.tensor(wts)
    npz('losses_extra', **out)


def golden_models_extra():
    """model_*.npz for the SURVEY 8f rank-4 plugins: the reference's LinearAEResidual(+Leaky) and Seq2SeqResidualA/B/C under
    a fixed state_dict, eval mode (BatchNorm running statistics perturbed so that they matter; Dropout off)."""
    if not os.path.isdir(REF_SRC):
        sys.exit('reference tree not present: the committed .npz files are the artefact to use')
    install_standins()
    sys.path.insert(0, REF_SRC)
    from pedestrians_video_2_carla.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla.modules.movements.linear_ae.linear_ae_residual import LinearAEResidual
    from pedestrians_video_2_carla.modules.movements.linear_ae.linear_ae_residual_leaky import LinearAEResidualLeaky
    from pedestrians_video_2_carla.modules.movements.seq2seq.seq2seq_residual_a import Seq2SeqResidualA
    from pedestrians_video_2_carla.modules.movements.seq2seq.seq2seq_residual_b import Seq2SeqResidualB
    from pedestrians_video_2_carla.modules.movements.seq2seq.seq2seq_residual_c import Seq2SeqResidualC
    g = torch.Generator().manual_seed(5)
    frames = torch.randn(4, 16, 26, 2, generator=g)
    for name, cls, kw in (
            ('linear_ae_residual', LinearAEResidual, {}),
            ('linear_ae_residual_leaky', LinearAEResidualLeaky, dict(linear_size=64)),
            # A and B: same architecture, seed and construction order as model_seq2seq_embeddings_pose_2d -> same weights;
            # only the outputs are stored (checked below). C (6-D output) gets a small configuration of its own.
            ('seq2seq_residual_a', Seq2SeqResidualA, dict(movements_output_type=MT.pose_2d)),
            ('seq2seq_residual_b', Seq2SeqResidualB, dict(movements_output_type=MT.pose_2d)),
            ('seq2seq_residual_c', Seq2SeqResidualC, dict(movements_output_type=MT.pose_changes, hidden_size=16,
                                                          single_joint_embeddings_size=8)),
    ):
        torch.manual_seed(22742)
        model = cls(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, **kw).eval()
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.3)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
        with torch.no_grad():
            out = model(frames)
        sd = {('sd__' + k): v for k, v in model.state_dict().items()}
        if name in ('seq2seq_residual_a', 'seq2seq_residual_b'):
            base = np.load(os.path.join(HERE, 'model_seq2seq_embeddings_pose_2d.npz'))
            assert all(np.array_equal(base[k], v.numpy()) for k, v in sd.items()) and len(sd) == sum(
                k.startswith('sd__') for k in base.files)
            sd = {}
        outs = {'out': out} if isinstance(out, torch.Tensor) else {'out_loc': out[0], 'out_rot': out[1]}
        npz('model_' + name, frames=frames, n_params=sum(p.numel() for p in model.parameters()), **outs, **sd)


def golden_seq2seq_h128():
    """model_seq2seq_embeddings_h128_pose_changes.npz: the reference's Seq2SeqEmbeddings at the hidden size its own configs use
    (configs/compare/carla-recorded_autoencoder_tests.yaml:38: hidden_size 128) with the pose_changes output (156 features per frame,
    seq2seq.py:245-288), eval mode, fixed state_dict. single_joint_embeddings_size=8 keeps the fixture at ~1.5 MB (the encoder's first
    input projection is 512 x (26 x 8) instead of 512 x 1664)."""
    if not os.path.isdir(REF_SRC):
        sys.exit('reference tree not present: the committed .npz files are the artefact to use')
    install_standins()
    sys.path.insert(0, REF_SRC)
    from pedestrians_video_2_carla.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla.modules.movements.seq2seq.seq2seq_embeddings import Seq2SeqEmbeddings
    g = torch.Generator().manual_seed(7)
    frames = torch.randn(3, 16, 26, 2, generator=g)
    torch.manual_seed(22742)
    model = Seq2SeqEmbeddings(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_changes,
                              hidden_size=128, single_joint_embeddings_size=8).eval()
    with torch.no_grad():
        out = model(frames)
    sd = {('sd__' + k): v for k, v in model.state_dict().items()}
    npz('model_seq2seq_embeddings_h128_pose_changes', frames=frames, out=out, n_params=sum(p.numel() for p in model.parameters()), **sd)


class StandInPoseTransformer(torch.nn.Module):
    """Stand-in for the THIRD-PARTY transformer ``third_party/PoseFormer common/model_poseformer.PoseTransformer`` (empty
    git submodule in the reference checkout, like pytorch3d a package the image lacks): same constructor keywords, same
    call contract -- (B, num_frame, J, in_chans) -> (B, 1, J, 3), the centre-frame pose -- computed by a small fixed map
    (a linear map of the flattened window followed by tanh, so that every frame and joint of the window matters and the
    order of the windows is visible in the result). It lets the REFERENCE's own wrapper ``PoseFormer.forward`` /
    ``eval_slice`` (pose_former.py:114-127) run unmodified; the transformer's arithmetic stays parity-unpinned."""

    def __init__(self, num_frame=9, num_joints=26, in_chans=2, **kwargs):
        super().__init__()
        self.num_joints = num_joints
        g = torch.Generator().manual_seed(97)
        self.map = torch.nn.Linear(num_frame * num_joints * in_chans, num_joints * 3)
        with torch.no_grad():
            self.map.weight.copy_(torch.randn(self.map.weight.shape, generator=g) * 0.05)
            self.map.bias.copy_(torch.randn(self.map.bias.shape, generator=g) * 0.1)

    def forward(self, x):
        return torch.tanh(self.map(x.reshape(x.shape[0], -1))).view(x.shape[0], 1, self.num_joints, 3)


def golden_wrappers():
    """Two behaviours the reference OWNS around third-party / framework code, run from the reference's own classes:
    (a) the PoseFormer window wrapper (pose_former.py:117-127) + eval_slice (114-115) for clip_length 30 and 81, with the
        stand-in above in place of the absent third-party transformer;
    (b) teacher forcing in Seq2Seq.forward / _decode_frame / _teacher_forcing (seq2seq.py:245-288, 323-349), train mode,
        Seq2SeqEmbeddings: frames_force and clip_force on pose_2d, frames_force on pose_changes (6-D targets through
        matrix_to_rotation_6d); dropout 0 so that the only random draw is the forcing decision, which is recorded as the
        uniform numbers torch.rand returned; output, a weighted-sum loss and every parameter gradient are stored."""
    if not os.path.isdir(REF_SRC):
        sys.exit('reference tree not present: the committed .npz files are the artefact to use')
    install_standins()
    sys.path.insert(0, REF_SRC)
    import pedestrians_video_2_carla  # noqa: F401
    import importlib
    # the stand-in is registered under the reference's import path of the third-party package (pose_former.py:6)
    for name in ('pedestrians_video_2_carla.third_party', 'pedestrians_video_2_carla.third_party.pose_former'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    tp = types.ModuleType('pedestrians_video_2_carla.third_party.pose_former.model_poseformer')
    tp.PoseTransformer = StandInPoseTransformer
    sys.modules[tp.__name__] = tp
    from pedestrians_video_2_carla.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla.modules.flow.output_types import MovementsModelOutputType as MT
    pf_mod = importlib.import_module('pedestrians_video_2_carla.modules.movements.pose_former.pose_former')
    assert pf_mod.PoseFormerModel is StandInPoseTransformer, 'the reference fell back to its NotAvailable dummy'
    out = {}
    for T in (30, 81):
        g = torch.Generator().manual_seed(200 + T)
        model = pf_mod.PoseFormer(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, clip_length=T).eval()
        x = torch.randn(3, T, 26, 2, generator=g)
        with torch.no_grad():
            y = model(x)
        sl = model.eval_slice
        out.update({f'T{T}_x': x, f'T{T}_out': y, f'T{T}_eval_slice': np.array([sl.start, sl.stop])})
        assert model.output_type == MT.absolute_loc
    inner = StandInPoseTransformer()
    npz('pose_former_wrapper', standin_weight=inner.map.weight, standin_bias=inner.map.bias, **out)

    from pedestrians_video_2_carla.modules.movements.seq2seq.seq2seq_embeddings import Seq2SeqEmbeddings
    from pytorch3d.transforms import euler_angles_to_matrix
    out = {}
    B, T = 5, 16
    small = dict(hidden_size=32, single_joint_embeddings_size=8)        # small models: the fixture stores weights + gradients
    for tag, kw in (('frames_pose_2d', dict(movements_output_type=MT.pose_2d, teacher_mode='frames_force', teacher_force_ratio=0.3, **small)),
                    ('clip_pose_2d', dict(movements_output_type=MT.pose_2d, teacher_mode='clip_force', teacher_force_ratio=0.4, **small)),
                    ('frames_pose_changes', dict(movements_output_type=MT.pose_changes, teacher_mode='frames_force',
                                                 teacher_force_ratio=0.3, hidden_size=16, single_joint_embeddings_size=8))):
        torch.manual_seed(22742)
        model = Seq2SeqEmbeddings(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, p_dropout=0.0, **kw).train()
        g = torch.Generator().manual_seed(31)
        x = torch.randn(B, T, 26, 2, generator=g)
        if kw['movements_output_type'] == MT.pose_2d:
            targets = {'projection_2d_transformed': torch.randn(B, T, 26, 2, generator=g)}
            w = torch.randn(B, T, 26, 2, generator=g)
        else:
            ang = (torch.rand(B, T, 26, 3, generator=g) * 2 - 1) * 0.3
            targets = {'pose_changes': euler_angles_to_matrix(ang, 'XYZ')}
            w = torch.randn(B, T, 26, 3, 3, generator=g)
        shape = (1, B) if kw['teacher_mode'] == 'clip_force' else (T, B)
        torch.manual_seed(777)
        uniform = torch.rand(shape)                       # what the reference's one torch.rand call returns after this seed
        torch.manual_seed(777)
        y = model(x, targets)
        loss = (y * w).sum()
        loss.backward()
        out.update({f'{tag}__x': x, f'{tag}__w': w, f'{tag}__uniform': uniform, f'{tag}__out': y, f'{tag}__loss': loss})
        out.update({f'{tag}__target__{k}': v for k, v in targets.items()})
        out.update({f'{tag}__sd__{k}': v for k, v in model.state_dict().items()})
        out.update({f'{tag}__grad__{k}': p.grad for k, p in model.named_parameters()})
        # the forced rows of the OUTPUT equal the targets (input and output are one tensor in _decode_frame)
        idx = (uniform < kw['teacher_force_ratio'])
        idx = idx.repeat(T, 1) if kw['teacher_mode'] == 'clip_force' else idx
        assert idx.any() and not idx.all()
        if kw['movements_output_type'] == MT.pose_2d:
            forced = y.permute(1, 0, 2, 3)[idx]
            assert torch.equal(forced, targets['projection_2d_transformed'].permute(1, 0, 2, 3)[idx])
    npz('teacher_forcing', **out)


def golden_metrics_extra():
    """Section 9b (the rest): the reference's own MultiinputWrapper (metrics/multiinput_wrapper.py) around a
    mean-squared-error base metric, and MissingJointsRatio (metrics/missing_joints_ratio.py), run on two batches each.
    torchmetrics is absent: ``Metric`` gets the stand-in of section 9 and ``MeanSquaredError`` its public definition
    (sum of squared errors / number of elements). The FB_* metrics wrap third_party/video_pose_3d (empty submodule): they
    cannot be run here and stay parity-unpinned."""
    class Metric(torch.nn.Module):
        def __init__(self, dist_sync_on_step=False, **kwargs):
            super().__init__()

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, default.clone())

    class MeanSquaredError(Metric):
        def __init__(self, **kwargs):
            super().__init__(**kwargs)
            self.add_state('sum_squared_error', torch.tensor(0.0))
            self.add_state('total', torch.tensor(0))

        def update(self, preds, target):
            self.sum_squared_error += ((preds - target) ** 2).sum()
            self.total += target.numel()

        def compute(self):
            return self.sum_squared_error / self.total

    _module('torchmetrics', Metric=Metric, MeanSquaredError=MeanSquaredError)
    from pedestrians_video_2_carla.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla.data.openpose.skeleton import BODY_25_SKELETON
    from pedestrians_video_2_carla.metrics.missing_joints_ratio import MissingJointsRatio
    from pedestrians_video_2_carla.metrics.multiinput_wrapper import MultiinputWrapper
    g = torch.Generator().manual_seed(101)
    out, batches = {}, []
    for k in range(2):
        B = 3 + k
        b = dict(pred=torch.randn(B, 16, 26, 2, generator=g), gt=torch.randn(B, 16, 26, 2, generator=g),
                 gt_b25=torch.randn(B, 16, 25, 2, generator=g))
        b['gt'][torch.rand(B, 16, 26, generator=g) < 0.15] = 0           # missing ground-truth joints
        b['gt_b25'][torch.rand(B, 16, 25, generator=g) < 0.15] = 0
        b['pred_mj'] = b['pred'].clone()
        b['pred_mj'][torch.rand(B, 16, 26, generator=g) < 0.2] = 0       # predicted "missing" joints for the ratio
        batches.append(b)
        for name, v in b.items():
            out[f'b{k}_{name}'] = v
    key = 'projection_2d_transformed'
    mse = MultiinputWrapper(MeanSquaredError(), key, key, input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON)
    mse_nomask = MultiinputWrapper(MeanSquaredError(), key, key, input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON,
                                   mask_missing_joints=False)
    mse_b25 = MultiinputWrapper(MeanSquaredError(), key, key, input_nodes=BODY_25_SKELETON, output_nodes=CARLA_SKELETON)
    mjr = MissingJointsRatio(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON)
    mjr_b25 = MissingJointsRatio(input_nodes=BODY_25_SKELETON, output_nodes=CARLA_SKELETON)
    for b in batches:
        mse.update({key: b['pred']}, {key: b['gt']})
        mse_nomask.update({key: b['pred']}, {key: b['gt']})
        mse_b25.update({key: b['pred']}, {key: b['gt_b25']})
        mjr.update({'projection_2d': b['pred_mj']}, {})
        mjr_b25.update({'projection_2d': b['pred_mj']}, {})
    npz('metrics_extra', mse=mse.compute(), mse_nomask=mse_nomask.compute(), mse_b25=mse_b25.compute(), mjr=mjr.compute(),
        mjr_b25=mjr_b25.compute(), mjr_present=mjr.present_joints, mjr_total=mjr.total, **out)


if __name__ == '__main__':
    if sys.argv[1:] == ['metrics_extra']:
        install_standins()
        sys.path.insert(0, REF_SRC)
        golden_metrics_extra()
    elif sys.argv[1:] == ['models_extra']:
        golden_models_extra()
    elif sys.argv[1:] == ['seq2seq_h128']:
        golden_seq2seq_h128()
    elif sys.argv[1:] == ['losses_extra']:
        golden_losses_extra()
    elif sys.argv[1:] == ['collate']:
        golden_collate()
    elif sys.argv[1:] == ['wrappers']:
        golden_wrappers()
    elif sys.argv[1:] == ['metrics']:
        install_standins()
        sys.path.insert(0, REF_SRC)
        golden_metrics()
    else:
        main()
        golden_metrics()
        golden_metrics_extra()
