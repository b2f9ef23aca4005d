"""Two behaviours the reference owns around code it does not own, checked against fixtures made by RUNNING the reference's
own classes (tests/golden/make_golden.py::golden_wrappers):

  * the PoseFormer window wrapper + eval_slice (reference modules/movements/pose_former/pose_former.py:114-127), with a small
    fixed module standing in for the absent third-party transformer (fixture ``pose_former_wrapper.npz``: its weights, inputs
    and the reference wrapper's outputs for clip_length 30 and 81);
  * teacher forcing in Seq2Seq.forward / _decode_frame / _teacher_forcing (seq2seq.py:245-288, 323-349), Seq2SeqEmbeddings in
    train mode (fixture ``teacher_forcing.npz``: state_dict, inputs, targets, the uniform numbers of the forcing draw,
    output, loss and every parameter gradient) -- including the reference's quirk that the forced rows are written INTO the
    decoder output, so they also are what the model returns.

CPU tests run the host path (fp64 and fp32); the GPU test runs the same modules on the device (HIP embeddings / LSTM ops)."""
import pytest
import torch


class StandIn(torch.nn.Module):
    """The fixture's stand-in transformer: tanh(linear(flattened window)) -> (B, 1, J, 3)."""

    def __init__(self, weight, bias):
        super().__init__()
        self.map = torch.nn.Linear(weight.shape[1], weight.shape[0])
        with torch.no_grad():
            self.map.weight.copy_(weight)
            self.map.bias.copy_(bias)

    def forward(self, x):
        return torch.tanh(self.map(x.reshape(x.shape[0], -1))).view(x.shape[0], 1, 26, 3)


def _pose_former(golden, T, device='cpu'):
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.movements.pose_former import PoseFormer
    g = golden('pose_former_wrapper')
    inner = StandIn(g['standin_weight'], g['standin_bias'])
    model = PoseFormer(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, clip_length=T, inner_model=inner).eval().to(device)
    return g, model


@pytest.mark.parametrize('T', [30, 81])
def test_pose_former_wrapper_equals_the_reference_wrapper(golden, T):
    g, model = _pose_former(golden, T)
    with torch.no_grad():
        y = model(g[f'T{T}_x'])
    want = g[f'T{T}_out']
    assert y.shape == want.shape == (3, T, 26, 3)
    assert torch.allclose(y, want, rtol=0, atol=5e-6), float((y - want).abs().max())   # one batched GEMM vs one per window
    assert (want[:, :4] == 0).all() and (y[:, :4] == 0).all()                 # frames without a full receptive field
    assert [model.eval_slice.start, model.eval_slice.stop] == g[f'T{T}_eval_slice'].tolist() == [4, T - 4]
    # the frames past the last window's centre repeat that window's prediction (the reference's broadcast write)
    assert torch.equal(want[:, T - 4:], want[:, T - 5:T - 4].expand(-1, 4, -1, -1))
    assert torch.equal(y[:, T - 4:], y[:, T - 5:T - 4].expand(-1, 4, -1, -1))


@pytest.mark.gpu
@pytest.mark.parametrize('T', [30, 81])
def test_pose_former_wrapper_on_the_device_equals_the_reference_wrapper(golden, T):
    assert torch.cuda.is_available()
    d = torch.device('cuda:0')
    g, model = _pose_former(golden, T, d)
    with torch.no_grad():
        y = model(g[f'T{T}_x'].to(d)).cpu()
    assert torch.allclose(y, g[f'T{T}_out'], rtol=0, atol=5e-6), float((y - g[f'T{T}_out']).abs().max())


CASES = {'frames_pose_2d': dict(otype='pose_2d', teacher_mode='frames_force', teacher_force_ratio=0.3, hidden_size=32,
                                single_joint_embeddings_size=8),
         'clip_pose_2d': dict(otype='pose_2d', teacher_mode='clip_force', teacher_force_ratio=0.4, hidden_size=32,
                              single_joint_embeddings_size=8),
         'frames_pose_changes': dict(otype='pose_changes', teacher_mode='frames_force', teacher_force_ratio=0.3, hidden_size=16,
                                     single_joint_embeddings_size=8)}


def _forcing_case(golden, tag, device, dtype):
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2SeqEmbeddings
    g = {k[len(tag) + 2:]: v for k, v in golden('teacher_forcing').items() if k.startswith(tag + '__')}
    kw = dict(CASES[tag])
    model = Seq2SeqEmbeddings(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, p_dropout=0.0,
                              movements_output_type=MT[kw.pop('otype')], **kw)
    model.load_state_dict({k[4:]: v for k, v in g.items() if k.startswith('sd__')})        # the reference's checkpoint keys
    model = model.to(device=device, dtype=dtype).train()
    targets = {k[8:]: v.to(device=device, dtype=dtype) for k, v in g.items() if k.startswith('target__')}
    return g, model, targets


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize('tag', list(CASES))
def test_teacher_forcing_equals_the_reference_on_the_host(golden, tag):
    """fp64 on the host; the forcing decision comes from the SAME torch.rand call after the same seed as in the reference run
    (dropout 0: no other draw), so nothing of _teacher_forcing is stubbed."""
    g, model, targets = _forcing_case(golden, tag, 'cpu', torch.float64)
    torch.manual_seed(777)
    y = model(g['x'].double(), targets)
    loss = (y * g['w'].double()).sum()
    loss.backward()
    assert _rel(y, g['out']) < 2e-6 and abs(float(loss) - float(g['loss'])) < 2e-5 * abs(float(g['loss']))
    for name, p in model.named_parameters():
        assert _rel(p.grad, g['grad__' + name]) < 5e-5, name                       # the fixture is the reference's fp32 run
    if 'projection_2d_transformed' in targets:                                   # forced rows of the output ARE the targets
        T, B = 16, g['x'].shape[0]
        idx = g['uniform'] < CASES[tag]['teacher_force_ratio']
        idx = idx.repeat(T, 1) if idx.shape[0] == 1 else idx
        assert torch.equal(y.permute(1, 0, 2, 3)[idx], targets['projection_2d_transformed'].permute(1, 0, 2, 3)[idx])


@pytest.mark.gpu
@pytest.mark.parametrize('tag', list(CASES))
def test_teacher_forcing_on_the_device_equals_the_reference(golden, tag, monkeypatch):
    """The same modules on the MI355X (folded HIP embeddings, HIP LSTM layer ops per decoded frame). The device generator is
    a different one, so torch.rand is made to return the reference's recorded uniform numbers; _teacher_forcing itself (the
    comparison with the ratio, clip_force's repeat over the frames, target formatting) runs unmodified."""
    assert torch.cuda.is_available()
    d = torch.device('cuda:0')
    g, model, targets = _forcing_case(golden, tag, d, torch.float32)
    real_rand = torch.rand
    calls = []

    def recorded(*size, device=None, **kw):
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        if shape == tuple(g['uniform'].shape):
            calls.append(shape)
            return g['uniform'].to(device)
        return real_rand(*size, device=device, **kw)
    monkeypatch.setattr(torch, 'rand', recorded)
    y = model(g['x'].to(d), targets)
    monkeypatch.setattr(torch, 'rand', real_rand)
    assert len(calls) == 1, 'exactly one forcing draw per forward'
    loss = (y * g['w'].to(d)).sum()
    loss.backward()
    assert _rel(y, g['out']) < 1e-4 and abs(float(loss) - float(g['loss'])) < 1e-4 * abs(float(g['loss']))
    for name, p in model.named_parameters():
        assert _rel(p.grad, g['grad__' + name]) < 2e-4, name
