"""GPU: the two-launch train step (csrc/p2c_train.hip, ops.fused_train_step) through the flow and the trainer.

Parity chain: oracle (fp64, CPU) <- separate kernels (tests/test_flow_gpu.py) <- fused step, the last link BITWISE at the
metric's batch size (same arithmetic, same summation orders), and the fused step directly against the oracle on ragged
shapes. Graph mode on NEW batches every step == eager (static batch staging)."""
import copy
import math
import os
import sys

import pytest
import torch

from oracle import pose_head as O

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_flow_gpu import close, dev, make  # noqa: E402


def _trainer(flow, dm, **kw):
    from pedestrians_video_2_carla_amd.trainer import Trainer
    return Trainer(device=dev(), **kw).setup(flow, dm)


def _took_fused_path(flow):
    return getattr(flow, '_pair_counts', None) is not None


@pytest.mark.parametrize('otype', ['pose_changes', 'relative_rot'])
def test_fused_step_is_bit_identical_to_the_separate_kernels(monkeypatch, otype):
    """B = 256 (one sample tile per CU: the separate path runs its split weight gradient): losses, parameters and both Adam
    moments after 5 steps are the same BITS with and without the fused step."""
    results = {}
    for fused in ('0', '1'):
        monkeypatch.setenv('P2C_FUSED_TRAIN', fused)
        flow, dm = make(B=256, otype=otype)
        trainer = _trainer(flow, dm)
        batch = dm.generate_batch(dev())
        losses = [trainer.train_step(flow, batch, i).clone() for i in range(5)]
        assert _took_fused_path(flow) == (fused == '1')
        opt = trainer.optimizers[0]
        st = opt.state[trainer.flat.flat_param]
        results[fused] = (torch.stack(losses).cpu(), trainer.flat.flat_param.detach().cpu().clone(),
                          st['exp_avg'].cpu().clone(), st['exp_avg_sq'].cpu().clone(), float(st['step']),
                          flow.logged['train_loss/loc_2d'].cpu().clone(), flow.logged['train_loss/loc_3d'].cpu().clone())
    a, b = results['0'], results['1']
    assert torch.isfinite(a[0]).all() and a[4] == b[4] == 5.0
    for i, what in enumerate(('losses', 'parameters', 'exp_avg', 'exp_avg_sq')):
        assert torch.equal(a[i], b[i]), f'{what}: max |diff| {(a[i] - b[i]).abs().max().item():.3e}'
    assert torch.equal(a[5], b[5]) and torch.equal(a[6], b[6])


@pytest.fixture
def first_launch_form(request):
    """'latency': train_clip_kernel (a workgroup per clip) at every batch size; 'stream': train_stream_kernel (a pair of
    wavefronts per clip, csrc/p2c_train_stream.hip) at every batch size; '+wgrad': the second launch in its throughput form
    (wgrad_stream_kernel + wgrad_reduce_kernel) at every batch size, else never. The library's own thresholds are restored."""
    from pedestrians_video_2_carla_amd import _lib
    lib = _lib.lib()
    prev = lib.p2c_train_step_set_stream_min_batch(1 if request.param.startswith('stream') else (1 << 30))
    prev_w = lib.p2c_train_step_set_wgrad_stream_min_batch(1 if request.param.endswith('wgrad') else (1 << 30))
    yield request.param
    lib.p2c_train_step_set_stream_min_batch(prev)
    lib.p2c_train_step_set_wgrad_stream_min_batch(prev_w)


@pytest.mark.parametrize('first_launch_form', ['latency', 'stream', 'stream+wgrad', 'latency+wgrad'], indirect=True)
@pytest.mark.parametrize('B,T,missing,transform,otype', [
    (5, 16, 0.1, 'hips_neck_bbox', 'pose_changes'), (33, 7, 0.2, 'hips_neck', 'pose_changes'),
    (64, 16, 0.0, 'bbox', 'pose_changes'), (1, 1, 0.0, 'none', 'pose_changes'),
    (300, 16, 0.1, 'hips_neck_bbox', 'pose_changes'), (777, 5, 0.1, 'hips_neck', 'pose_changes'),
    (40, 16, 0.1, 'hips_neck', 'relative_rot'), (1030, 16, 0.1, 'hips_neck', 'pose_changes'), (19, 15, 0.3, 'none', 'relative_rot')])
def test_fused_step_matches_cpu_pipeline(monkeypatch, first_launch_form, B, T, missing, transform, otype):
    """loss + every parameter gradient of one fused train step == LinearAE on CPU (fp64) + oracle pose head; ragged batch,
    clips shorter than the 16-sample tile, every built-in transform, missing joints, both 6-D kinds; B > 256: the per-clip
    kernel is persistent (a workgroup walks 2 - 4 clips, weight image staged once); both forms of the first launch."""
    from pedestrians_video_2_carla_amd.data.base.base_transforms import BaseTransforms
    monkeypatch.setenv('P2C_FUSED_UPDATE', '0')                  # keep the gradients: the optimizer is a separate launch
    flow, dm = make(B=B, T=T, missing=missing, transform=BaseTransforms[transform], otype=otype)
    kind = {'pose_changes': 'pose_changes_6d', 'relative_rot': 'relative_rot_6d'}[otype]
    cpu_model = copy.deepcopy(flow.movements_model).double()
    trainer = _trainer(flow, dm)
    trainer.optimizers[0].zero_grad_in_step = False
    batch = dm.generate_batch(dev())
    frames, targets, meta = batch
    loss = trainer._forward_backward(flow, batch, 0)
    torch.cuda.synchronize()
    assert _took_fused_path(flow)
    gt2d = targets['projection_2d_transformed' if transform != 'none' else 'projection_2d']
    o = O.pose_head(cpu_model(frames.double().cpu()), kind, meta['skel_type'].cpu(), transform=transform,
                    gt2d=gt2d.double().cpu(), gt3d=targets['absolute_pose_loc'].double().cpu())
    o['loc_2d_3d'].backward()
    close(loss, o['loc_2d_3d'], 'loss')
    close(flow.logged['train_loss/loc_2d'], o['loc_2d'], 'loc_2d')
    close(flow.logged['train_loss/loc_3d'], o['loc_3d'], 'loc_3d')
    # fp32 tolerance rule of tests/test_pose_head_gpu.py: max(1e-4, 2 x the error the fp32 CPU pipeline itself makes)
    cpu32 = copy.deepcopy(cpu_model).float()
    o32 = O.pose_head(cpu32(frames.float().cpu()), kind, meta['skel_type'].cpu(), transform=transform,
                      gt2d=gt2d.float().cpu(), gt3d=targets['absolute_pose_loc'].float().cpu())
    o32['loc_2d_3d'].backward()
    for (n, p), q, q32 in zip(flow.movements_model.named_parameters(), cpu_model.parameters(), cpu32.parameters()):
        ref_err = (q32.grad.double() - q.grad).abs().max().item() / (q.grad.abs().max().item() + 1e-30)
        close(p.grad, q.grad, n, rtol=max(1e-4, 2 * ref_err))


@pytest.mark.parametrize('B', [1024, 8192])
def test_throughput_forms_match_the_separate_kernels_at_full_size(monkeypatch, B):
    """B = 1 024 (cfg2 / cfg4 per GPU) and 8 192, 10 % missing joints, default thresholds (train_stream_kernel; at 8 192 also the
    streamed weight gradient): loss and the whole flat gradient of one step against the separate kernels (mlp_fwd / pose head /
    mlp_bwd: P2C_FUSED_TRAIN=0), 1e-4 of the gradient's scale; the loss itself to 1e-6."""
    grads = {}
    monkeypatch.setenv('P2C_FUSED_UPDATE', '0')
    for fused in ('0', '1'):
        monkeypatch.setenv('P2C_FUSED_TRAIN', fused)
        flow, dm = make(B=B, missing=0.1)
        trainer = _trainer(flow, dm)
        trainer.optimizers[0].zero_grad_in_step = False
        batch = dm.generate_batch(dev())
        loss = trainer._forward_backward(flow, batch, 0)
        torch.cuda.synchronize()
        assert _took_fused_path(flow) == (fused == '1')
        grads[fused] = (loss.clone(), trainer.flat.flat_grad.clone())
    close(grads['1'][0], grads['0'][0], 'loss', rtol=1e-6)
    close(grads['1'][1], grads['0'][1], 'flat gradient', rtol=1e-4)


@pytest.mark.parametrize('first_launch_form', ['stream+wgrad', 'latency'], indirect=True)
def test_eval_slice_in_both_forms_matches_the_separate_kernels(monkeypatch, first_launch_form):
    """An eval slice (frames 2 .. 12 of 16 carry the losses, the rotations still accumulate over all frames): both forms of the fused
    step against the separate kernels, B = 44 with 10 % missing joints."""
    grads = {}
    monkeypatch.setenv('P2C_FUSED_UPDATE', '0')
    for fused in ('0', '1'):
        monkeypatch.setenv('P2C_FUSED_TRAIN', fused)
        flow, dm = make(B=44, missing=0.1)
        monkeypatch.setattr(type(flow.movements_model), 'eval_slice', property(lambda self: slice(2, 13)))
        trainer = _trainer(flow, dm)
        trainer.optimizers[0].zero_grad_in_step = False
        batch = dm.generate_batch(dev())
        loss = trainer._forward_backward(flow, batch, 0)
        assert _took_fused_path(flow) == (fused == '1')
        grads[fused] = (loss.clone(), trainer.flat.flat_grad.clone())
    close(grads['1'][0], grads['0'][0], 'loss', rtol=1e-6)
    close(grads['1'][1], grads['0'][1], 'flat gradient', rtol=1e-4)


def test_fused_step_with_world_motion_and_body25_targets_matches_separate_kernels(monkeypatch):
    """Non-identity trajectory (dloc / drot) and an eval slice: fused step vs the separate kernels (tolerance: the two paths
    run different weight-gradient orders at this batch size)."""
    from pedestrians_video_2_carla_amd.modules.trajectory.trajectory import TrajectoryModel
    from pedestrians_video_2_carla_amd.modules.flow.output_types import TrajectoryModelOutputType

    class Drift(TrajectoryModel):
        output_type = property(lambda self: TrajectoryModelOutputType.changes)

        def configure_optimizers(self):
            return {}

        def forward(self, x, *a, **k):
            B, T = x.shape[:2]
            g = torch.Generator().manual_seed(3)
            dloc = (torch.randn(B, T, 3, generator=g) * 0.01).to(x.device)
            ang = (torch.randn(B, T, generator=g) * 0.02).to(x.device)
            c, s = ang.cos(), ang.sin()
            z, o = torch.zeros_like(c), torch.ones_like(c)
            drot = torch.stack((c, -s, z, s, c, z, z, z, o), -1).view(B, T, 3, 3)
            return dloc, drot

    grads = {}
    monkeypatch.setenv('P2C_FUSED_UPDATE', '0')
    for fused in ('0', '1'):
        monkeypatch.setenv('P2C_FUSED_TRAIN', fused)
        flow, dm = make(B=24, missing=0.1)
        flow.trajectory_model = Drift()
        monkeypatch.setattr(type(flow.movements_model), 'eval_slice', property(lambda self: slice(2, 13)))
        trainer = _trainer(flow, dm)
        trainer.optimizers[0].zero_grad_in_step = False
        batch = dm.generate_batch(dev())
        loss = trainer._forward_backward(flow, batch, 0)
        assert _took_fused_path(flow) == (fused == '1')
        grads[fused] = (loss.clone(), trainer.flat.flat_grad.clone())
    close(grads['1'][0], grads['0'][0], 'loss', rtol=1e-6)
    close(grads['1'][1], grads['0'][1], 'flat gradient', rtol=2e-5)


@pytest.mark.parametrize('fused', ['0', '1'])
def test_graph_mode_trains_on_new_batches_like_eager(monkeypatch, fused):
    """Trainer(use_graph=True) fed a DIFFERENT batch every step == eager on the same 20 batches (static batch staging);
    a resident batch object is not re-staged."""
    monkeypatch.setenv('P2C_FUSED_TRAIN', fused)
    curves = {}
    for graph in (False, True):
        flow, dm = make(B=32, missing=0.1)
        trainer = _trainer(flow, dm, use_graph=graph)
        losses = []
        for i, batch in enumerate(dm.train_batches(dev(), 20)):
            batch[2].pop('skel_type')                                    # the loader's meta: lists of strings only
            losses.append(trainer.train_step(flow, batch, i).clone())
        curves[graph] = torch.stack(losses).cpu()
        if graph:
            staged = trainer._static_batch
            before = staged[0].clone()
            trainer.train_step(flow, batch, 20)                          # same object again: nothing copied
            assert torch.equal(staged[0], before) and trainer._staged_src is batch
    assert torch.isfinite(curves[True]).all()
    assert len(set(curves[False].tolist())) > 15                          # the batches really differ
    assert torch.equal(curves[True], curves[False]), (curves[True] - curves[False]).abs().max()


@pytest.mark.parametrize('direct', ['1', '0'])
def test_captured_step_replayed_as_its_recorded_call(monkeypatch, direct):
    """Single GPU, two-launch step with the optimizer inside: the capture records the one p2c_train_step call, the captured graph
    is checked to hold exactly its two kernel nodes, and the step is then replayed by making that call directly (no graph
    start-up). Same losses, bit for bit, as eager and as the graph replay (P2C_DIRECT_REPLAY=0), across a learning-rate change;
    a model the fused step does not cover keeps the graph."""
    monkeypatch.setenv('P2C_DIRECT_REPLAY', direct)
    flow, dm = make(B=64, missing=0.1)
    eager_flow, _ = make(B=64, missing=0.1)
    eager = _trainer(eager_flow, dm)
    trainer = _trainer(flow, dm, use_graph=True)
    batches = list(dm.train_batches(dev(), 12))
    def run(tr, fl):
        out = []
        for i, b in enumerate(batches):
            if i == 6:                                                    # an LR-scheduler step between replays reaches the launch
                for grp in tr.optimizers[0].param_groups:
                    grp['lr'] *= 0.1
            out.append(tr.train_step(fl, b, i).clone())
        return torch.stack(out).cpu()
    got, want = run(trainer, flow), run(eager, eager_flow)
    assert _took_fused_path(flow)
    assert (trainer._direct is not None) == (direct == '1')
    assert torch.equal(got, want), (got - want).abs().max()
    monkeypatch.setenv('P2C_FUSED_TRAIN', '0')                           # separate kernels: seven launches, the graph stays
    flow2, dm2 = make(B=64, missing=0.1)
    t2 = _trainer(flow2, dm2, use_graph=True)
    t2.train_step(flow2, dm2.generate_batch(dev()), 0)
    assert t2._direct is None


def test_graph_mode_rejects_a_batch_of_another_structure():
    flow, dm = make(B=8)
    trainer = _trainer(flow, dm, use_graph=True)
    trainer.train_step(flow, dm.generate_batch(dev()), 0)
    other = dm.generate_batch(dev(), batch_size=4)
    with pytest.raises(RuntimeError, match='fixed structure'):
        trainer.train_step(flow, other, 1)


def test_fused_step_errors_and_count_kernel():
    from pedestrians_video_2_carla_amd import _lib, ops
    d = dev()
    g = torch.Generator().manual_seed(5)
    gt2d = torch.randn(6, 9, 26, 2, generator=g)
    gt2d[torch.rand(6, 9, 26, generator=g) < 0.3] = 0.0
    spec = ops.PoseHeadSpec(kind='pose_changes_6d', eval_slice=(1, 8))
    counts = ops.count_target_pairs(spec, gt2d.to(d)).cpu()
    mask = (gt2d != 0).all(-1)
    mask[..., 1] = True                                                    # hips are never masked (tensors.py:33-38)
    assert torch.equal(counts, mask[:, 1:8].sum((1, 2)).float())
    with pytest.raises(_lib.P2CError):                                     # outside the deferred context
        ops.fused_train_step(torch.zeros(2, 16, 26, 2, device=d), [], [], spec, torch.zeros(2, dtype=torch.int32, device=d),
                             torch.zeros(2, device=d))
    assert not ops.train_step_supported([52, 26, 13, 6, 19, 39, 78], 16) and not ops.train_step_supported(ops.LINEAR_AE_6D_DIMS, 17)


def test_optimizer_step_is_not_skipped_when_the_fused_backward_did_not_run():
    """with the optimizer riding on the backward, a step whose module took another path (eval mode) must still be
    applied -- by the stand-alone optimizer launch."""
    flow, dm = make(B=16)
    trainer = _trainer(flow, dm)
    assert trainer._opt_in_backward
    batch = dm.generate_batch(dev())
    trainer.train_step(flow, batch, 0)
    flow.movements_model.eval()                                            # LinearAE.fused_args: no fused optimizer in eval()
    before = trainer.flat.flat_param.detach().clone()
    trainer.train_step(flow, batch, 1)
    torch.cuda.synchronize()
    assert float(trainer.optimizers[0].state[trainer.flat.flat_param]['step']) == 2.0
    assert not torch.equal(before, trainer.flat.flat_param.detach())


def test_direct_replay_takes_new_batches_by_address():
    """Direct replay (the captured step is one recorded C-ABI call): a new batch with the static batch's layout is not copied --
    the call's four input addresses are pointed at its tensors and its target pairs counted -- and a batch that does not have
    that layout (strings instead of the skeleton-type index; non-contiguous frames) goes through the copying path with the
    addresses set back to the static buffers. Same losses, bit for bit, as eager on the same sequence of batches."""
    flow, dm = make(B=48, missing=0.1)
    eager_flow, _ = make(B=48, missing=0.1)
    eager = _trainer(eager_flow, dm)
    trainer = _trainer(flow, dm, use_graph=True)
    batches = list(dm.train_batches(dev(), 9))
    batches[3][2].pop('skel_type')                                       # -> copying path (meta of strings only)
    wide = torch.zeros(48, 16, 26, 4, device=dev())
    wide[..., :2] = batches[6][0]
    batches[6] = (wide[..., :2], batches[6][1], batches[6][2])           # non-contiguous frames -> copying path
    got, want, how = [], [], []
    for i, b in enumerate(batches):
        static_before = trainer._static_batch[0].clone() if trainer._static_batch is not None else None
        got.append(trainer.train_step(flow, b, i).clone())
        eb = (b[0].contiguous(), b[1], dict(b[2]))
        want.append(eager.train_step(eager_flow, eb, i).clone())
        how.append('handed over' if trainer._handed_over is b else 'copied')
        if how[-1] == 'handed over':
            assert torch.equal(trainer._static_batch[0], static_before), 'a handed-over batch is not copied'
            assert trainer._direct['desc'].mlp.x == b[0].data_ptr()
        elif i > 0:
            assert trainer._direct['desc'].mlp.x == trainer._static_batch[0].data_ptr()
    assert trainer._direct is not None
    assert how == ['copied', 'handed over', 'handed over', 'copied', 'handed over', 'handed over', 'copied', 'handed over',
                   'handed over'], how
    assert torch.equal(torch.stack(got), torch.stack(want)), (torch.stack(got) - torch.stack(want)).abs().max()
