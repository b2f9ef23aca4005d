"""GPU: PoseFormer (BASELINE.json configs[4]) through the pose-lifting flow at clip_length 81.

The transformer's arithmetic is parity-unpinned (third-party source absent from the reference checkout): what is checked is
what the reference owns -- the window wrapper (pose_former.py:117-127), eval_slice (:114-115), the optimizer / scheduler
(:129-138) -- plus properties of the build's restatement (shapes, centre-frame dependence, permutation of the batch) and
one captured train step of cfg5's per-GPU share."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _model(T):
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.movements.pose_former import PoseFormer
    torch.manual_seed(5)
    return PoseFormer(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, clip_length=T)


def test_batched_windows_equal_the_reference_loop_and_properties():
    T = 30
    m = _model(T).to(dev()).eval()
    assert sum(p.numel() for p in m.parameters()) == 22298744      # 4 + 4 blocks at widths 32 / 832, 26 joints
    x = torch.randn(3, T, 26, 2, device=dev())
    with torch.no_grad():
        y = m(x)
        ref = torch.zeros(3, T, 26, 3, device=dev())
        for i in range(T - 9 + 1):                                 # pose_former.py:121-125, written out
            ref[:, i + 4:i + 9 + 4] = m.pose_former(x[:, i:i + 9])
        assert torch.allclose(y, ref, rtol=1e-4, atol=1e-5), (y - ref).abs().max()
        assert m.eval_slice == slice(4, 26) and (y[:, :4] == 0).all()
        # a window's output depends on its nine frames only, and clips do not talk to each other
        x2 = x.clone()
        x2[:, 20:] += 1.0
        y2 = m(x2)
        assert torch.equal(y2[:, 4:15], y[:, 4:15]) and not torch.allclose(y2[:, 16:], y[:, 16:])
        assert torch.allclose(m(x[[2, 0, 1]]), y[[2, 0, 1]], rtol=1e-4, atol=1e-5)
    cfg = m.configure_optimizers()
    assert cfg['optimizer'].defaults['lr'] == 4e-4 and cfg['optimizer'].defaults['weight_decay'] == 0.1
    assert isinstance(cfg['lr_scheduler'], torch.optim.lr_scheduler.ExponentialLR) and cfg['lr_scheduler'].gamma == 0.99


def test_cfg5_train_steps_through_the_flow_with_lr_schedule():
    """clip_length 81, BODY_25 data mapped onto the CARLA input skeleton with zero fill, absolute_loc head (K1b), loss over
    frames [4, 77), AdamW + ExponentialLR stepped at epoch ends; eager and captured steps agree."""
    from pedestrians_video_2_carla_amd.data.base.skeleton import get_common_indices
    from pedestrians_video_2_carla_amd.data.carla.carla_recorded_synthetic import SyntheticCarlaRecordedDataModule
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.data.openpose.skeleton import BODY_25_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.pose_lifting import LitPoseLiftingFlow
    from pedestrians_video_2_carla_amd.trainer import Trainer
    d, T, B = dev(), 81, 8
    dm = SyntheticCarlaRecordedDataModule(clip_length=T, batch_size=B)
    frames, targets, meta = dm.generate_batch(d)
    carla_idx, _ = get_common_indices(input_nodes=BODY_25_SKELETON, output_nodes=CARLA_SKELETON)
    missing = [j for j in range(26) if j not in carla_idx]         # CARLA joints OpenPose does not see: zero-filled
    frames[:, :, missing] = 0
    targets['projection_2d_transformed'][:, :, missing] = 0
    batch = (frames, targets, meta)
    curves = {}
    for graph in (False, True):
        flow = LitPoseLiftingFlow(movements_model=_model(T), loss_modes=['loc_2d_3d'], transform='hips_neck_bbox')
        trainer = Trainer(device=d, use_graph=graph, steps_per_epoch=2).setup(flow, dm)
        assert len(trainer.lr_schedulers) == 1 and trainer.lr_schedulers[0]['interval'] == 'epoch'
        flow.movements_model.train()
        for blk in list(flow.movements_model.pose_former.Spatial_blocks) + list(flow.movements_model.pose_former.blocks):
            blk.drop_path.p = 0.0                                   # stochastic depth off: the two runs must agree
        losses = trainer.fit_steps(flow, [batch] * 4) if hasattr(trainer, 'fit_steps') else None
        if losses is None:
            losses = []
            for i in range(4):
                losses.append(trainer.train_step(flow, batch, i).clone())
                if (i + 1) % 2 == 0:
                    trainer.current_epoch += 1
                    trainer.step_lr_schedulers('epoch')
        curves[graph] = torch.stack(losses).cpu()
        lr = trainer.optimizers[0].param_groups[0]['lr']
        assert abs(lr - 4e-4 * 0.99 ** 2) < 1e-12, lr               # two epoch ends
    if os.environ.get('P2C_PRINT_CURVES'):
        print('curves', curves)
    assert torch.isfinite(curves[False]).all() and curves[False][-1] < curves[False][0]
    assert torch.allclose(curves[True], curves[False], rtol=2e-3), (curves[True], curves[False])


@pytest.mark.gpu
@pytest.mark.parametrize('S,N,heads,hd', [(37, 26, 8, 4), (11, 9, 8, 104), (5, 1, 2, 6), (3, 17, 3, 12), (6, 16, 2, 16), (4, 12, 4, 8),
                                          (2, 33, 2, 12), (3, 40, 1, 4), (300, 5, 3, 20)])
def test_small_attention_matches_fp64(S, N, heads, hd):
    """K14 (p2c_attn_small_fwd/_bwd) on PoseTransformer's two shapes (26 joint tokens x 8 heads x 4, 9 frame tokens x 8 x 104) and
    on every kernel family and row width (narrow 4 / 8, matrix-core wide, generic vector and scalar; N <= 16 / 32 / 64) against softmax(scale q k^T) v written out in fp64: output and the gradient of qkv, 1e-5 relative."""
    import torch
    from pedestrians_video_2_carla_amd import ops
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(S * 31 + N)
    qkv64 = torch.randn(S, N, 3, heads, hd, generator=g, dtype=torch.float64, requires_grad=True)
    up = torch.randn(S, N, heads * hd, generator=g, dtype=torch.float64)
    scale = hd ** -0.5
    q, k, v = qkv64.permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(S, N, heads * hd)
    (ref * up).sum().backward()
    qkv = qkv64.detach().float().to(d).requires_grad_(True)
    out = ops.small_attention(qkv, scale)
    (out * up.float().to(d)).sum().backward()

    def close(a, b, what):
        a, b = a.detach().double().cpu(), b.detach()
        err, sc = (a - b).abs().max().item(), b.abs().max().item()
        assert err <= 1e-5 * sc, f'{what}: {err:.3e} vs scale {sc:.3e}'
    close(out, ref, 'out'), close(qkv.grad, qkv64.grad, 'grad qkv')


@pytest.mark.gpu
@pytest.mark.parametrize('rows,D,sinks', [(5000, 32, False), (777, 832, True), (3, 4, False), (1025, 64, False), (130, 100, True),
                                           (64, 1024, False), (100, 260, False)])
def test_layer_norm_matches_fp64(rows, D, sinks):
    """K15 (p2c_layernorm_fwd/_bwd) against torch.nn.functional.layer_norm in fp64 on PoseTransformer's two widths and
    every lane grouping of the kernel: output, input gradient, gamma / beta gradients (returned, or added into .grad = ones)."""
    import torch
    from pedestrians_video_2_carla_amd import ops
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(rows + D)
    x64 = (torch.randn(rows, D, generator=g, dtype=torch.float64) * 2 + 0.5).requires_grad_(True)
    w64 = torch.randn(D, generator=g, dtype=torch.float64).requires_grad_(True)
    b64 = torch.randn(D, generator=g, dtype=torch.float64).requires_grad_(True)
    up = torch.randn(rows, D, generator=g, dtype=torch.float64)
    ref = torch.nn.functional.layer_norm(x64, (D,), w64, b64, 1e-6)
    (ref * up).sum().backward()
    x = x64.detach().float().to(d).requires_grad_(True)
    w = torch.nn.Parameter(w64.detach().float().to(d))
    b = torch.nn.Parameter(b64.detach().float().to(d))
    if sinks:
        w.grad, b.grad = torch.ones_like(w), torch.ones_like(b)
    with ops.grad_sinks(sinks):
        y = ops.layer_norm(x.view(1, rows, D), w, b, 1e-6).view(rows, D)
        (y * up.float().to(d)).sum().backward()

    def close(a, ref_, what, rtol=2e-5):
        a, ref_ = a.detach().double().cpu(), ref_.detach()
        err, sc = (a - ref_).abs().max().item(), ref_.abs().max().item()
        assert err <= rtol * sc, f'{what}: {err:.3e} vs scale {sc:.3e}'
    close(y, ref, 'y'), close(x.grad, x64.grad, 'grad x', 5e-5)
    close(w.grad, w64.grad + (1 if sinks else 0), 'grad gamma', 5e-5 * max(1.0, rows ** 0.5))
    close(b.grad, b64.grad + (1 if sinks else 0), 'grad beta', 5e-5 * max(1.0, rows ** 0.5))


@pytest.mark.gpu
def test_spatial_half_once_per_frame_equals_the_per_window_model():
    """PoseFormer(share_spatial): the per-frame half of the transformer run on T frames instead of on every (window, frame)
    pair. Eval mode: same outputs as the per-window evaluation (the default takes the shared path by itself there). Training
    without stochastic depth: same loss gradient w.r.t. every parameter. With stochastic depth the default keeps the
    per-window evaluation."""
    import torch
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.movements.pose_former import PoseFormer
    d = torch.device('cuda:0')
    torch.manual_seed(4)
    kw = dict(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, clip_length=20, drop_path_rate=0.0)
    model = PoseFormer(**kw).to(d)
    x = torch.randn(3, 20, 26, 2, device=d)
    up = torch.randn(3, 20, 26, 3, device=d)

    def run(share, train):
        model.share_spatial = share
        model.train(train)
        model.zero_grad()
        y = model(x)
        (y * up).sum().backward()
        return y.detach(), {n: p.grad.clone() for n, p in model.named_parameters()}

    y0, g0 = run(False, False)
    y1, g1 = run(None, False)             # default in eval mode: shared
    assert model.pose_former.spatial_is_deterministic()
    torch.testing.assert_close(y1, y0, rtol=1e-4, atol=1e-5)
    y2, g2 = run(None, True)              # training, no stochastic depth: still the same function -> shared
    y3, g3 = run(False, True)
    torch.testing.assert_close(y2, y3, rtol=1e-4, atol=1e-5)
    for n in g2:
        scale = g3[n].abs().max().item() + 1e-12
        assert (g2[n] - g3[n]).abs().max().item() <= 2e-4 * scale, n
    stochastic = PoseFormer(**{**kw, 'drop_path_rate': 0.2}).to(d).train()
    assert not stochastic.pose_former.spatial_is_deterministic()          # default: per-window evaluation in training
    stochastic.eval()
    assert stochastic.pose_former.spatial_is_deterministic()


@pytest.mark.gpu
@pytest.mark.parametrize('S,N,C,heads,p_drop,sinks', [(40, 26, 32, 8, 0.3, False), (12, 9, 832, 8, 0.25, True), (7, 9, 160, 8, 0.0, False),
                                                      (2600, 26, 32, 8, 0.2, True)])     # (67 600 rows: the long-row dW routing)
def test_block_as_one_autograd_node_matches_the_same_block_in_fp64(S, N, C, heads, p_drop, sinks):
    """pose_transformer._Block through ops.transformer_block (one node: LayerNorm / GEMM / attention launches with the factor,
    residual, GELU and bias work in their epilogues, LayerNorm backward adding the residual gradient, parameter gradients added
    into the sinks) against the block's written-out formula in fp64 with the SAME stochastic-depth draws: output, input gradient
    and all 12 parameter gradients -- with fresh gradients and accumulated into existing ones inside ``grad_sinks``."""
    import copy
    import math
    from pedestrians_video_2_carla_amd import ops
    from pedestrians_video_2_carla_amd.modules.movements.pose_former.pose_transformer import _Block
    d = dev()
    torch.manual_seed(S + C)
    blk = _Block(C, heads, 2.0, True, None, 0.0, 0.0, p_drop, lambda n: torch.nn.LayerNorm(n, eps=1e-6)).to(d).train()
    for prm in blk.parameters():
        prm.data.add_(torch.randn_like(prm) * 0.05)
    ref = copy.deepcopy(blk).double()
    x = torch.randn(S, N, C, device=d, requires_grad=True)
    up = torch.randn(S, N, C, device=d)
    seeds = []
    if sinks:
        for prm in blk.parameters():
            prm.grad = torch.randn_like(prm)
            seeds.append(prm.grad.clone())
    assert blk._one_node(x)
    torch.manual_seed(99)
    with ops.grad_sinks(sinks):
        y = blk(x)
        (y * up).sum().backward()
    # fp64 formula with the same two draws per sample
    torch.manual_seed(99)
    keep = 1.0 - p_drop
    f1 = f2 = None
    if p_drop > 0:
        f1 = x.new_empty(S).bernoulli_(keep).div_(keep).double()
        f2 = x.new_empty(S).bernoulli_(keep).div_(keep).double()
    x64 = x.detach().double().requires_grad_(True)
    hd = C // heads

    def attention(t):
        qkv = ref.attn.qkv(t).reshape(S, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
        w = torch.softmax((qkv[0] @ qkv[1].transpose(-1, -2)) * ref.attn.scale, -1)
        return ref.attn.proj((w @ qkv[2]).transpose(1, 2).reshape(S, N, C))

    def gelu(t):
        return 0.5 * t * (1 + torch.erf(t / math.sqrt(2.0)))
    br1 = attention(ref.norm1(x64))
    x1 = x64 + (br1 if f1 is None else br1 * f1.view(-1, 1, 1))
    br2 = ref.mlp.fc2(gelu(ref.mlp.fc1(ref.norm2(x1))))
    y64 = x1 + (br2 if f2 is None else br2 * f2.view(-1, 1, 1))
    (y64 * up.double()).sum().backward()

    def rel(a, b):
        return float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
    assert rel(y, y64) < 2e-5
    assert rel(x.grad, x64.grad) < 1e-4
    for i, ((name, pg), (_, pr)) in enumerate(zip(blk.named_parameters(), ref.named_parameters())):
        want = pr.grad + (seeds[i].double() if sinks else 0.0)
        assert rel(pg.grad, want) < 1e-4, name


@pytest.mark.gpu
@pytest.mark.parametrize('B,F,C', [(2336, 9, 832), (5, 3, 8), (1, 1, 4), (100, 64, 36)])
def test_frame_mean_and_row_parameter_ops_match_fp64(B, F, C):
    """ops.frame_mean (p2c_frame_mean_fwd + K12 contractions behind it) and ops.add_row_parameter (K12 column sum behind it)
    against the written-out formulas in fp64: values and all gradients."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    torch.manual_seed(B + F)
    x = torch.randn(B, F, C, device=d, requires_grad=True)
    w, b = torch.randn(F, 1, device=d, requires_grad=True), torch.randn(1, device=d, requires_grad=True)
    p = torch.randn(1, F, C, device=d, requires_grad=True)
    up = torch.randn(B, C, device=d)
    y = ops.frame_mean(ops.add_row_parameter(x, p), w.view(1, F, 1), b)
    (y * up).sum().backward()
    x64, w64, b64, p64 = (t.detach().double().requires_grad_(True) for t in (x, w, b, p))
    y64 = ((x64 + p64) * w64.view(1, F, 1)).sum(1) + b64
    (y64 * up.double()).sum().backward()

    def rel(a, r):
        return float((a.detach().double() - r).abs().max() / (r.abs().max() + 1e-30))
    assert rel(y, y64) < 1e-5
    for name, a, r in (('x', x, x64), ('w', w, w64), ('b', b, b64), ('p', p, p64)):
        assert rel(a.grad, r.grad) < 5e-5, name
