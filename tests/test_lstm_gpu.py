"""LSTM layer = library GEMM + HIP recurrence (K7b, p2c_lstm_rec_fwd/_bwd through the C ABI) against torch.nn.LSTM in
fp64 on the CPU with the same weights: outputs, final states and every gradient within 1e-4 relative."""
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(autouse=True, params=['narrow', 'wide'])
def rec_tile(request, monkeypatch):
    """Every case runs on both tilings of the time-loop kernels: 4 sequences per workgroup (v_mfma_f32_4x4x1_16B, the default up
    to B = 4096) and 16 (v_mfma_f32_16x16x4). The library reads the switch at each call."""
    monkeypatch.setenv('P2C_REC_TILE', request.param)
    return request.param


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def close(a, b, what, rtol=RTOL):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err, scale = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= rtol * scale + 1e-30, f'{what}: {err:.3e} vs scale {scale:.3e}'


@pytest.mark.parametrize('T,B,I,H', [(1, 1, 5, 16), (16, 6, 52, 64), (7, 33, 20, 32), (16, 130, 64, 64), (3, 17, 9, 48),
                                     (16, 37, 52, 128), (5, 130, 156, 128), (9, 21, 20, 96)])      # 128: the reference's own hidden_size (configs/compare/carla-recorded_autoencoder_tests.yaml:38)
@pytest.mark.parametrize('with_state', [False, True])
def test_layer_matches_torch_lstm(T, B, I, H, with_state):
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    torch.manual_seed(T * 100 + B)
    ref = torch.nn.LSTM(I, H).double()
    x = torch.randn(T, B, I, dtype=torch.float64)
    h0 = torch.randn(B, H, dtype=torch.float64) if with_state else torch.zeros(B, H, dtype=torch.float64)
    c0 = torch.randn(B, H, dtype=torch.float64) if with_state else torch.zeros(B, H, dtype=torch.float64)
    up, uh, uc = torch.randn(T, B, H, dtype=torch.float64), torch.randn(B, H, dtype=torch.float64), torch.randn(B, H, dtype=torch.float64)
    xr, hr, cr = x.clone().requires_grad_(True), h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
    out_r, (hT_r, cT_r) = ref(xr, (hr[None], cr[None]))
    ((out_r * up).sum() + (hT_r[0] * uh).sum() + (cT_r[0] * uc).sum()).backward()

    p = {n: v.detach().float().to(d).requires_grad_(True) for n, v in ref.named_parameters()}
    xd, hd, cd = (t.float().to(d).requires_grad_(True) for t in (x, h0, c0))
    out, hT, cT = ops.lstm_layer(xd, hd, cd, p['weight_ih_l0'], p['weight_hh_l0'], p['bias_ih_l0'], p['bias_hh_l0'])
    ((out * up.float().to(d)).sum() + (hT * uh.float().to(d)).sum() + (cT * uc.float().to(d)).sum()).backward()
    close(out, out_r, 'out'), close(hT, hT_r[0], 'hT'), close(cT, cT_r[0], 'cT')
    close(xd.grad, xr.grad, 'grad x'), close(hd.grad, hr.grad, 'grad h0'), close(cd.grad, cr.grad, 'grad c0')
    for n, v in ref.named_parameters():
        close(p[n].grad, v.grad, 'grad ' + n)


def test_seq2seq_model_uses_the_fused_stack_and_matches_cpu():
    """Whole Seq2SeqEmbeddings model (encoder stack, T decoder steps, fc) on the GPU vs the same module in fp64 on the CPU
    (nn.LSTM path): forward and all parameter gradients."""
    import copy
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2SeqEmbeddings
    d = dev()
    torch.manual_seed(3)
    model = Seq2SeqEmbeddings(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_2d,
                              p_dropout=0.0).train()
    cpu = copy.deepcopy(model).double()
    gpu = model.to(d)
    x = torch.randn(9, 16, 26, 2)
    up = torch.randn(9, 16, 26, 2)
    cpu32 = copy.deepcopy(cpu).float()                 # the reference's own fp32 arithmetic (nn.LSTM on the CPU)
    yr = cpu(x.double())
    (yr * up.double()).sum().backward()
    y32 = cpu32(x)
    (y32 * up).sum().backward()
    y = gpu(x.to(d))
    (y * up.to(d)).sum().backward()

    def bound(a32, a64):       # max(1e-4, 2 x the error fp32 on the CPU makes against fp64): tests/test_pose_head_gpu.py's rule
        return max(1e-4, 2.0 * (a32.double() - a64).abs().max().item() / (a64.abs().max().item() + 1e-30))
    close(y, yr, 'model output', rtol=bound(y32, yr))
    for (n, pg), (_, pc), (_, p32) in zip(gpu.named_parameters(), cpu.named_parameters(), cpu32.named_parameters()):
        close(pg.grad, pc.grad, 'grad ' + n, rtol=bound(p32.grad, pc.grad))


@pytest.mark.parametrize('hidden,otype,O', [(128, 'pose_2d', 2), (128, 'pose_changes', 6), (96, 'pose_2d', 2)])
def test_seq2seq_shapes_of_the_reference_configs_stay_on_the_hip_recurrence(hidden, otype, O, monkeypatch):
    """hidden_size 128 (reference configs/compare/carla-recorded_autoencoder_tests.yaml:38,45,48) and the pose_changes output
    (156 features per frame, reference seq2seq.py:245-288: the decoder loop takes the per-step path through K7b): whole
    Seq2SeqEmbeddings model against the same module in fp64 on the CPU; nn.LSTM.forward must not be entered on the GPU side and no
    fall-back warning may be raised."""
    import copy
    import warnings
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2SeqEmbeddings
    d = dev()
    torch.manual_seed(5)
    model = Seq2SeqEmbeddings(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT[otype], p_dropout=0.0,
                              hidden_size=hidden).train()
    cpu = copy.deepcopy(model).double()
    cpu32 = copy.deepcopy(model)
    x = torch.randn(5, 16, 26, 2)
    yr = cpu(x.double())
    up = torch.randn(*yr.shape)
    (yr * up.double()).sum().backward()
    y32 = cpu32(x)
    (y32 * up).sum().backward()
    gpu = model.to(d)

    def refuse(*a, **k):
        raise AssertionError('nn.LSTM.forward entered: the stack left the HIP path')
    monkeypatch.setattr(torch.nn.LSTM, 'forward', refuse)
    with warnings.catch_warnings():
        warnings.simplefilter('error', RuntimeWarning)
        y = gpu(x.to(d))
        (y * up.to(d)).sum().backward()
    monkeypatch.undo()

    def bound(a32, a64):
        return max(1e-4, 2.0 * (a32.double() - a64).abs().max().item() / (a64.abs().max().item() + 1e-30))
    close(y, yr, 'model output', rtol=bound(y32, yr))
    for (n, pg), (_, pc), (_, p32) in zip(gpu.named_parameters(), cpu.named_parameters(), cpu32.named_parameters()):
        assert pg.grad is not None, n
        close(pg.grad, pc.grad, 'grad ' + n, rtol=bound(p32.grad, pc.grad))


def test_reference_run_of_the_hidden_128_model_on_the_hip_recurrence(golden, monkeypatch):
    """tests/golden/model_seq2seq_embeddings_h128_pose_changes.npz = the REFERENCE's Seq2SeqEmbeddings(hidden_size=128, pose_changes) run by
    tests/golden/make_golden.py: its state_dict on the device, its frames in, its output back within 1e-4 -- through the HIP recurrence
    (nn.LSTM.forward refused)."""
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2SeqEmbeddings
    g = golden('model_seq2seq_embeddings_h128_pose_changes')
    model = Seq2SeqEmbeddings(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_changes,
                              hidden_size=128, single_joint_embeddings_size=8).eval()
    model.load_state_dict({k[4:]: v for k, v in g.items() if k.startswith('sd__')})
    model = model.to(dev())

    def refuse(*a, **k):
        raise AssertionError('nn.LSTM.forward entered: the stack left the HIP path')
    monkeypatch.setattr(torch.nn.LSTM, 'forward', refuse)
    with torch.no_grad():
        out = model(g['frames'].to(dev()))
    monkeypatch.undo()
    close(out, g['out'].double(), 'reference output', rtol=1e-4)


@pytest.mark.parametrize('embeddings', [False, True])
def test_bidirectional_seq2seq_runs_on_the_hip_recurrence_and_matches_cpu(embeddings, monkeypatch):
    """bidirectional=True (reference seq2seq.py:36-38,72-73: encoder AND decoder stacks are bidirectional, fc reads 2H): the
    reverse direction is the same HIP recurrence over the time-flipped sequence. Whole model against the same module in fp64 on
    the CPU (nn.LSTM); nn.LSTM.forward must not be entered on the GPU side."""
    import copy
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2Seq, Seq2SeqEmbeddings
    d = dev()
    torch.manual_seed(11)
    cls = Seq2SeqEmbeddings if embeddings else Seq2Seq
    model = cls(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_2d, p_dropout=0.0,
                bidirectional=True, hidden_size=32).train()
    cpu = copy.deepcopy(model).double()
    cpu32 = copy.deepcopy(model)
    x, up = torch.randn(7, 6, 26, 2), torch.randn(7, 6, 26, 2)
    yr = cpu(x.double())
    (yr * up.double()).sum().backward()
    y32 = cpu32(x)
    (y32 * up).sum().backward()
    gpu = model.to(d)

    def refuse(*a, **k):
        raise AssertionError('nn.LSTM.forward entered: the bidirectional stack left the HIP path')
    monkeypatch.setattr(torch.nn.LSTM, 'forward', refuse)
    y = gpu(x.to(d))
    (y * up.to(d)).sum().backward()
    monkeypatch.undo()

    def bound(a32, a64):
        return max(1e-4, 2.0 * (a32.double() - a64).abs().max().item() / (a64.abs().max().item() + 1e-30))
    close(y, yr, 'model output', rtol=bound(y32, yr))
    for (n, pg), (_, pc), (_, p32) in zip(gpu.named_parameters(), cpu.named_parameters(), cpu32.named_parameters()):
        assert pg.grad is not None, n
        close(pg.grad, pc.grad, 'grad ' + n, rtol=bound(p32.grad, pc.grad))


@pytest.mark.parametrize('tile', ['narrow', 'wide'])
@pytest.mark.parametrize('mode', ['frames_force', 'clip_force'])
def test_teacher_forcing_inside_the_decoder_launch(tile, mode, monkeypatch):
    """Teacher forcing (reference seq2seq.py:272-288,323-349) in K7c: forced frames are replaced by their targets inside the one
    decoder launch (output AND next input, no gradient through them). Against the per-step path of the same model (K7b layer
    launches + torch.where, the path tests/test_reference_wrappers.py pins to the reference's own run) with the same draws:
    output and every parameter gradient."""
    import copy
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2Seq
    d = dev()
    monkeypatch.setenv('P2C_REC_TILE', tile)
    torch.manual_seed(17)
    model = Seq2Seq(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_2d, p_dropout=0.0,
                    teacher_mode=mode, teacher_force_ratio=0.4).to(d).train()
    twin = copy.deepcopy(model)
    B, T = 21, 12
    x, up = torch.randn(B, T, 26, 2, device=d), torch.randn(B, T, 26, 2, device=d)
    targets = {'projection_2d_transformed': torch.randn(B, T, 26, 2, device=d)}
    assert model._decoder_loop_fusable(x)
    torch.manual_seed(5)
    y = model(x, targets)
    (y * up).sum().backward()
    monkeypatch.setattr(type(twin), '_decoder_loop_fusable', lambda self, x: False)
    torch.manual_seed(5)
    y_ref = twin(x, targets)
    (y_ref * up).sum().backward()
    monkeypatch.undo()
    forced = (y == targets['projection_2d_transformed']).all(-1).all(-1)           # (B,T) frames that are their targets
    assert 0.2 < float(forced.float().mean()) < 0.6
    if mode == 'clip_force':
        assert bool((forced.all(1) | (~forced).all(1)).all())
    close(y, y_ref, 'forced decoder output', rtol=2e-5)
    for (n, p), (_, q) in zip(model.named_parameters(), twin.named_parameters()):
        close(p.grad, q.grad, 'grad ' + n, rtol=2e-4)


def test_no_cpu_fallback():
    from pedestrians_video_2_carla_amd import ops, _lib
    with pytest.raises(_lib.P2CError):
        ops.lstm_layer(torch.zeros(2, 3, 4), torch.zeros(3, 16), torch.zeros(3, 16), torch.zeros(64, 4), torch.zeros(64, 16),
                       None, None)


@pytest.mark.parametrize('T,B,O,with_drop', [(1, 1, 52, False), (16, 37, 52, True), (5, 16, 64, False), (16, 130, 12, True)])
def test_decoder_loop_matches_the_per_step_formula(T, B, O, with_drop):
    """K7c against the reference loop written out in fp64: out_t = fc(cell1(cell0(x_t))) with the frozen encoder state."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    H = 64
    g = torch.Generator().manual_seed(T * 7 + B)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64) * 0.3
    P = {'k0': rnd(B, 4 * H), 'c0': rnd(B, H), 'k1': rnd(B, 4 * H), 'c1': rnd(B, H), 'w_ih0': rnd(4 * H, O),
         'w_ih1': rnd(4 * H, H), 'w_fc': rnd(O, H), 'b_fc': rnd(O)}
    drop = (torch.rand(T, B, H, generator=g) < 0.8).double() / 0.8 if with_drop else None
    up = torch.randn(T, B, O, generator=g, dtype=torch.float64)

    def cell(gates, c_enc):
        i, f, gg, o = gates.chunk(4, -1)
        return torch.sigmoid(o) * torch.tanh(torch.sigmoid(f) * c_enc + torch.sigmoid(i) * torch.tanh(gg))

    R = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    x, outs = torch.zeros(B, O, dtype=torch.float64), []
    for t in range(T):
        h0 = cell(x @ R['w_ih0'].t() + R['k0'], R['c0'])
        if drop is not None:
            h0 = h0 * drop[t]
        h1 = cell(h0 @ R['w_ih1'].t() + R['k1'], R['c1'])
        x = h1 @ R['w_fc'].t() + R['b_fc']
        outs.append(x)
    ref = torch.stack(outs)
    (ref * up).sum().backward()

    D = {k: v.float().to(d).requires_grad_(True) for k, v in P.items()}
    out = ops.decoder_loop(D['k0'], D['c0'], D['k1'], D['c1'], D['w_ih0'], D['w_ih1'], D['w_fc'], D['b_fc'], T,
                           None if drop is None else drop.float().to(d))
    (out * up.float().to(d)).sum().backward()
    close(out, ref, 'out')
    for k in P:
        close(D[k].grad, R[k].grad, 'grad ' + k, rtol=2e-4)


@pytest.mark.parametrize('T,B,O,with_drop,sinks', [(1, 1, 52, False, False), (16, 37, 52, True, True), (5, 16, 64, False, False),
                                                    (16, 130, 12, True, False)])
def test_decoder_stack_from_the_encoder_state(T, B, O, with_drop, sinks):
    """The decoder in one launch each way WITH its frame-invariant terms (k_l = b_ih_l + b_hh_l + hidden_l W_hh_l^T formed in
    the kernel, d k_l / d hidden_l in the backward, output and its gradient batch-first) against the per-frame formula of
    the reference in fp64: output, d hidden, d cell and all ten parameter gradients -- returned to autograd, or added into
    existing .grad tensors inside ``grad_sinks`` (the flat trainer's mode; they start from ones here)."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    H = 64
    g = torch.Generator().manual_seed(T * 11 + B)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64) * 0.3
    dec64 = torch.nn.LSTM(O, H, num_layers=2).double()
    fc64 = torch.nn.Linear(H, O).double()
    hidden64, cell64 = rnd(2, B, H).requires_grad_(True), rnd(2, B, H).requires_grad_(True)
    drop = (torch.rand(T, B, H, generator=g) < 0.8).double() / 0.8 if with_drop else None
    up = torch.randn(B, T, O, generator=g, dtype=torch.float64)

    def cell(gates, c_enc):
        i, f, gg, o = gates.chunk(4, -1)
        return torch.sigmoid(o) * torch.tanh(torch.sigmoid(f) * c_enc + torch.sigmoid(i) * torch.tanh(gg))

    k0 = hidden64[0] @ dec64.weight_hh_l0.t() + dec64.bias_ih_l0 + dec64.bias_hh_l0
    k1 = hidden64[1] @ dec64.weight_hh_l1.t() + dec64.bias_ih_l1 + dec64.bias_hh_l1
    x, outs = torch.zeros(B, O, dtype=torch.float64), []
    for t in range(T):
        h0 = cell(x @ dec64.weight_ih_l0.t() + k0, cell64[0])
        if drop is not None:
            h0 = h0 * drop[t]
        h1 = cell(h0 @ dec64.weight_ih_l1.t() + k1, cell64[1])
        x = fc64(h1)
        outs.append(x)
    ref = torch.stack(outs, 1)                                        # (B,T,O)
    (ref * up).sum().backward()

    dec = torch.nn.LSTM(O, H, num_layers=2).to(d)
    fc = torch.nn.Linear(H, O).to(d)
    dec.load_state_dict({k: v.float() for k, v in dec64.state_dict().items()})
    fc.load_state_dict({k: v.float() for k, v in fc64.state_dict().items()})
    params = list(dec.parameters()) + list(fc.parameters())
    if sinks:
        for p_ in params:
            p_.grad = torch.ones_like(p_)
    hidden, cellg = hidden64.detach().float().to(d).requires_grad_(True), cell64.detach().float().to(d).requires_grad_(True)
    with ops.grad_sinks(sinks):
        out = ops.decoder_stack(hidden, cellg, dec, fc, T, None if drop is None else drop.float().to(d))
        (out * up.float().to(d)).sum().backward()
    assert out.shape == (B, T, O)
    close(out, ref, 'out')
    close(hidden.grad, hidden64.grad, 'grad hidden', rtol=2e-4), close(cellg.grad, cell64.grad, 'grad cell', rtol=2e-4)
    for (name, p_), q in zip(list(dec.named_parameters()) + list(fc.named_parameters()), list(dec64.parameters()) + list(fc64.parameters())):
        want = q.grad + 1 if sinks else q.grad
        close(p_.grad, want, 'grad ' + name, rtol=2e-4)


@pytest.mark.parametrize('T,B,I,H,sinks', [(1, 3, 5, 16, False), (16, 37, 52, 64, True), (9, 130, 20, 32, False), (16, 64, 52, 64, False),
                                           (16, 37, 52, 128, True), (7, 19, 20, 96, False)])
def test_encoder_stack_matches_torch_lstm(T, B, I, H, sinks):
    """The 2-layer encoder as one explicit launch sequence (batch-first input, biases added in the recurrence, final states
    written into the stacked tensors, one grouped weight-gradient launch) against torch.nn.LSTM(num_layers=2) in fp64: final
    hidden / cell of both layers and all eight parameter gradients (returned to autograd, or added into .grad = ones)."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    torch.manual_seed(T * 3 + B)
    ref = torch.nn.LSTM(I, H, num_layers=2).double()
    x64 = torch.randn(B, T, I, dtype=torch.float64)
    up_h, up_c = torch.randn(2, B, H, dtype=torch.float64), torch.randn(2, B, H, dtype=torch.float64)
    _, (h64, c64) = ref(x64.transpose(0, 1))
    ((h64 * up_h).sum() + (c64 * up_c).sum()).backward()

    rnn = torch.nn.LSTM(I, H, num_layers=2).to(d)
    rnn.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    if sinks:
        for p_ in rnn.parameters():
            p_.grad = torch.ones_like(p_)
    with ops.grad_sinks(sinks):
        hidden, cell = ops.encoder_stack(x64.float().to(d), rnn)
        ((hidden * up_h.float().to(d)).sum() + (cell * up_c.float().to(d)).sum()).backward()
    close(hidden, h64, 'hidden'), close(cell, c64, 'cell')
    for (name, p_), q in zip(rnn.named_parameters(), ref.parameters()):
        close(p_.grad, q.grad + 1 if sinks else q.grad, 'grad ' + name, rtol=2e-4)
    # only the cell state used downstream: the unused output's gradient stays None inside the function
    rnn.zero_grad()
    hidden, cell = ops.encoder_stack(x64.float().to(d), rnn)
    (cell * up_c.float().to(d)).sum().backward()
    ref.zero_grad()
    _, (h64, c64) = ref(x64.transpose(0, 1))
    (c64 * up_c).sum().backward()
    close(rnn.weight_ih_l0.grad, ref.weight_ih_l0.grad, 'grad weight_ih_l0 (cell only)', rtol=2e-4)


def test_encoder_stack_inter_layer_dropout():
    """Training mode with nn.LSTM's inter-layer dropout: the explicit sequence draws the mask with torch's generator (same
    seed -> same result) and its backward is the derivative of that forward (directional finite difference in fp32)."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    torch.manual_seed(3)
    rnn = torch.nn.LSTM(20, 32, num_layers=2, dropout=0.3).to(d).train()
    x = torch.randn(33, 7, 20, device=d)
    up = torch.randn(2, 33, 32, device=d)

    def run(seed, w=None):
        torch.manual_seed(seed)
        if w is not None:
            with torch.no_grad():
                rnn.weight_ih_l0.copy_(w)
        hidden, cell = ops.encoder_stack(x, rnn)
        return ((hidden + cell) * up).sum()

    rnn.zero_grad()
    a = run(11)
    a.backward()
    g = rnn.weight_ih_l0.grad.clone()
    assert torch.equal(run(11), a) and not torch.equal(run(12), a)
    w0, direction = rnn.weight_ih_l0.detach().clone(), torch.randn_like(rnn.weight_ih_l0)
    eps = 1e-2
    with torch.no_grad():
        fd = (run(11, w0 + eps * direction) - run(11, w0 - eps * direction)) / (2 * eps)
        run(11, w0)
    assert abs(fd.item() - (g * direction).sum().item()) <= 2e-2 * max(1.0, abs(fd.item()))


def _hash_mask(state, site, p, shape):
    """numpy restatement of csrc/p2c_rec_dev.h (drop_begin / drop_value) for the FORWARD of the step `state` is at."""
    import numpy as np
    M = 0xFFFFFFFF

    def mix(x):
        x = x.astype(np.uint64)
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(0x7feb352d)) & np.uint64(M)
        x ^= x >> np.uint64(15)
        x = (x * np.uint64(0x846ca68b)) & np.uint64(M)
        x ^= x >> np.uint64(16)
        return x
    s0, s1, step = (int(v) & M for v in state[:3])
    k0 = int(mix(np.array([s0 ^ ((step * 0x9E3779B9) & M) ^ (((site + 1) * 0x632BE59B) & M)], dtype=np.uint64))[0])
    k1 = int(mix(np.array([(s1 + step + 0x85EBCA6B * (site + 1)) & M], dtype=np.uint64))[0])
    n = 1
    for v in shape:
        n *= v
    e = np.arange(n, dtype=np.uint64)
    h = mix((e * np.uint64(0x9E3779B1) + np.uint64(k0)) & np.uint64(M)) ^ np.uint64(k1)
    keep = h >= np.uint64(int(p * 4294967296.0))
    return torch.from_numpy((keep.astype(np.float32) * np.float32(1.0 / (1.0 - p))).reshape(shape))


@pytest.mark.parametrize('T,B,H', [(16, 37, 64), (5, 130, 32), (9, 21, 128)])
def test_dropout_drawn_inside_the_recurrence(T, B, H):
    """p2c_lstm_desc.out_drop / drop_state: the forward writes out * mask beside the raw output, mask = the documented hash of
    (seed, step, site, element) -- restated in numpy -- on both tilings; it leaves state = {.., step, step + 1}; the backward
    takes g_out as the gradient of the dropped output (= the plain backward fed g_out * mask) and leaves
    {.., step + 1, step + 1}; the keep rate is 1 - p."""
    import ctypes
    from pedestrians_video_2_carla_amd import _lib, ops
    d = dev()
    lib = _lib.lib()
    p, site = 0.2, 3
    g = torch.Generator(device=d).manual_seed(T + B + H)
    gx, w_hh = torch.randn(T, B, 4 * H, device=d, generator=g) * 0.5, torch.randn(4 * H, H, device=d, generator=g) * 0.2
    g_out = torch.randn(T, B, H, device=d, generator=g)
    state = ops.dropout_state(d)
    state[2:] = torch.tensor([7, 0], dtype=torch.int32)
    before = state.cpu().tolist()
    f = dict(dtype=torch.float32, device=d)
    out, out_drop, acts, cs = torch.empty(T, B, H, **f), torch.empty(T, B, H, **f), torch.empty(T, B, 4 * H, **f), torch.empty(T, B, H, **f)

    def desc(hashed):
        q = _lib.LstmDesc()
        q.T, q.B, q.H = T, B, H
        q.gx, q.w_hh, q.out, q.acts, q.cs = gx.data_ptr(), w_hh.data_ptr(), out.data_ptr(), acts.data_ptr(), cs.data_ptr()
        if hashed:
            q.out_drop, q.drop_state, q.drop_p, q.drop_site = out_drop.data_ptr(), state.data_ptr(), p, site
        return q
    _lib.check(lib.p2c_lstm_rec_fwd(ctypes.byref(desc(True)), ops._stream()), 'fwd')
    torch.cuda.synchronize()
    assert state.cpu().tolist() == before[:3] + [8]
    mask = _hash_mask(before, site, p, (T, B, H)).to(d)
    assert torch.equal(out_drop, out * mask)
    assert abs(float((mask > 0).float().mean()) - (1 - p)) < 4 * (p * (1 - p) / mask.numel()) ** 0.5
    g_plain, g_hash = torch.empty(T, B, 4 * H, **f), torch.empty(T, B, 4 * H, **f)
    q = desc(False)
    fed = g_out * mask
    q.g_out, q.g_gx = fed.data_ptr(), g_plain.data_ptr()
    _lib.check(lib.p2c_lstm_rec_bwd(ctypes.byref(q), ops._stream()), 'bwd')
    q = desc(True)
    q.g_out, q.g_gx = g_out.data_ptr(), g_hash.data_ptr()
    _lib.check(lib.p2c_lstm_rec_bwd(ctypes.byref(q), ops._stream()), 'bwd')
    torch.cuda.synchronize()
    close(g_hash, g_plain, 'd gates', rtol=1e-6)           # (the kernel's g_out * mask may contract into the next fma)
    assert state.cpu().tolist() == before[:2] + [8, 8]


@pytest.mark.parametrize('T,B,O', [(16, 37, 52), (5, 130, 12)])
def test_decoder_dropout_drawn_inside_the_kernels(T, B, O):
    """ops.decoder_stack with drop = (state, p, site): output and every gradient equal, bit for bit, the run that READS the same
    mask as a tensor (the numpy restatement of the hash), on both tilings; two steps in a row draw different masks."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    H, p, site = 64, 0.2, 1
    torch.manual_seed(T * 3 + B)
    dec, fc = torch.nn.LSTM(O, H, num_layers=2).to(d), torch.nn.Linear(H, O).to(d)
    hidden0, cell0 = torch.randn(2, B, H, device=d) * 0.3, torch.randn(2, B, H, device=d) * 0.3
    up = torch.randn(B, T, O, device=d)
    state = ops.dropout_state(d)
    start = state.cpu().tolist()

    def run(drop):
        for q in list(dec.parameters()) + list(fc.parameters()):
            q.grad = None
        hidden, cell = hidden0.clone().requires_grad_(True), cell0.clone().requires_grad_(True)
        out = ops.decoder_stack(hidden, cell, dec, fc, T, drop)
        (out * up).sum().backward()
        return [out.detach().clone(), hidden.grad.clone(), cell.grad.clone()] + [q.grad.clone() for q in list(dec.parameters()) + list(fc.parameters())]
    got = run((state, p, site))
    torch.cuda.synchronize()
    assert state.cpu().tolist() == start[:2] + [1, 1]
    want = run(_hash_mask(start, site, p, (T, B, H)).to(d))
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), i
    again = run((state, p, site))                          # the next step of the stream: another mask
    assert not torch.equal(again[0], got[0])
    want2 = run(_hash_mask(start[:2] + [1, 1], site, p, (T, B, H)).to(d))
    assert torch.equal(again[0], want2[0])


@pytest.mark.parametrize('T,B,I,H', [(16, 37, 52, 64), (7, 19, 20, 32)])
def test_encoder_stack_with_dropout_drawn_inside_the_recurrence(T, B, I, H):
    """ops.encoder_stack(drop_state=...): final states and all parameter gradients against torch.nn.LSTM layers in fp64 with the
    SAME inter-layer mask (the numpy restatement of the hash, site 0), 1e-4."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    p = 0.2
    torch.manual_seed(T + B + I)
    rnn = torch.nn.LSTM(I, H, num_layers=2, dropout=p, batch_first=True).to(d).train()
    x = torch.randn(B, T, I, device=d)
    up_h, up_c = torch.randn(2, B, H, device=d), torch.randn(2, B, H, device=d)
    state = ops.dropout_state(d)
    start = state.cpu().tolist()
    hidden, cell = ops.encoder_stack(x, rnn, drop_state=state)
    ((hidden * up_h).sum() + (cell * up_c).sum()).backward()
    mask = _hash_mask(start, 0, p, (T, B, H)).double()
    l0, l1 = torch.nn.LSTM(I, H).double(), torch.nn.LSTM(H, H).double()
    sd = {k: v.detach().double().cpu() for k, v in rnn.state_dict().items()}
    l0.load_state_dict({k[:-1] + '0': sd[k] for k in sd if k.endswith('_l0')})
    l1.load_state_dict({k[:-1] + '0': sd[k] for k in sd if k.endswith('_l1')})
    o0, (h0, c0) = l0(x.double().cpu().transpose(0, 1))
    o1, (h1, c1) = l1(o0 * mask)
    ((torch.stack([h0[0], h1[0]]) * up_h.double().cpu()).sum() + (torch.stack([c0[0], c1[0]]) * up_c.double().cpu()).sum()).backward()
    close(hidden, torch.stack([h0[0], h1[0]]), 'hidden'), close(cell, torch.stack([c0[0], c1[0]]), 'cell')
    for name, q in rnn.named_parameters():
        ref = (l0 if name.endswith('_l0') else l1)
        close(q.grad, getattr(ref, name[:-1] + '0').grad, 'grad ' + name, rtol=5e-4)


def test_atb_group_equals_the_single_launches():
    """p2c_atb_group: five problems of different shapes (strided rows, bias, second bias destination, accumulate) behind one
    launch pair give bit for bit what p2c_atb gives one by one."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    torch.manual_seed(5)
    shapes = [(8192 - 512, 256, 52), (8192, 256, 64), (8192, 52, 64), (512, 256, 64), (300, 7, 3)]
    A = [torch.randn(K, M, device=d) for K, M, N in shapes]
    Bm = [torch.randn(K, N + 3, device=d)[:, 1:1 + N] for K, M, N in shapes]
    single = [ops.atb(a, b, bias=(i >= 2)) for i, (a, b) in enumerate(zip(A, Bm))]
    acc0 = torch.randn(256, 52, device=d)
    want0 = acc0.clone()
    ops.atb(A[0], Bm[0], out=want0, accumulate=True)
    got0, b3, b3b = acc0.clone(), torch.zeros(256, device=d), torch.zeros(256, device=d)
    res = ops.atb_group([dict(a=A[0], b=Bm[0], out=got0, accumulate=True), dict(a=A[1], b=Bm[1]),
                         dict(a=A[2], b=Bm[2], bias=True), dict(a=A[3], b=Bm[3], bias=True, bias_out=b3, bias_out2=b3b),
                         dict(a=A[4], b=Bm[4], bias=True)])
    assert torch.equal(got0, want0)
    for i in (1, 2, 3, 4):
        assert torch.equal(res[i][0], single[i][0]), i
    assert torch.equal(res[2][1], single[2][1]) and torch.equal(res[4][1], single[4][1])
    assert torch.equal(b3, single[3][1]) and torch.equal(b3b, single[3][1])


def test_folded_embeddings_with_the_flat_trainer_match_the_unfolded_model():
    """Seq2SeqEmbeddings under the flat trainer (embedding parameters / gradients addressed as strided blocks of the flat
    buffers, ``_FoldedInputMap``): loss and the whole flat gradient of one train step vs the same model with the fold
    switched off (K7a embeddings + the 1664-wide input projection)."""
    from pedestrians_video_2_carla_amd.data.carla.carla_recorded_synthetic import SyntheticCarlaRecordedDataModule
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.autoencoder import LitAutoencoderFlow
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2SeqEmbeddings
    from pedestrians_video_2_carla_amd.modules.movements.seq2seq import seq2seq_embeddings as SE
    from pedestrians_video_2_carla_amd.trainer import Trainer, seed_everything
    d = dev()
    res = {}
    for fold in (True, False):
        seed_everything(11)
        dm = SyntheticCarlaRecordedDataModule(clip_length=16, batch_size=24)
        model = Seq2SeqEmbeddings(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_2d,
                                  p_dropout=0.0)
        model.fold_embeddings = fold
        flow = LitAutoencoderFlow(movements_model=model, loss_modes=['loc_2d'], transform='hips_neck_bbox')
        trainer = Trainer(device=d, use_graph=False).setup(flow, dm)
        assert model.grad_sink
        if fold:      # the strided-block path is the one that runs
            ws = [e.weight for e in model.embeddings]
            assert SE._block_view(ws) is not None and SE._block_view([p.grad for p in ws]) is not None
        batch = dm.generate_batch(d)
        flow.train()
        flow.on_train_batch_start(batch, 0)
        loss = flow.training_step(batch, 0)['loss']
        loss.backward()
        res[fold] = (loss.detach().clone(), trainer.flat.flat_grad.detach().clone())
    close(res[True][0], res[False][0], 'loss', rtol=1e-5)
    close(res[True][1], res[False][1], 'flat gradient', rtol=2e-4)


@pytest.mark.parametrize('variant', ['A', 'B'])
def test_residual_seq2seq_variants_on_the_gpu_match_cpu(variant):
    """Seq2SeqResidualA/B (per-frame decoder through the fused LSTM layer op, folded encoder) vs the module in fp64 on the
    CPU: output and every parameter gradient, with teacher forcing on identical forcing decisions."""
    import copy
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.modules.movements import seq2seq
    d = dev()
    torch.manual_seed(4)
    cls = getattr(seq2seq, 'Seq2SeqResidual' + variant)
    model = cls(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_2d, p_dropout=0.0,
                teacher_mode='frames_force', teacher_force_ratio=0.3).train()
    cpu, gpu = copy.deepcopy(model).double(), model.to(d)
    x, up, tgt = torch.randn(7, 16, 26, 2), torch.randn(7, 16, 26, 2), torch.randn(7, 16, 26, 2)
    idx = torch.rand(16, 7) < 0.3
    for m, dev_, dt in ((cpu, 'cpu', torch.float64), (gpu, d, torch.float32)):
        forced = tgt.permute(1, 0, 2, 3).reshape(16, 7, 52).to(dev_, dt)
        m._teacher_forcing = lambda targets, f=forced, i=idx.to(dev_): (True, f, i)
    yr = cpu(x.double(), {})
    (yr * up.double()).sum().backward()
    y = gpu(x.to(d), {})
    (y * up.to(d)).sum().backward()
    close(y, yr, 'model output', rtol=2e-4)
    for (n, pg), (_, pc) in zip(gpu.named_parameters(), cpu.named_parameters()):
        close(pg.grad, pc.grad, 'grad ' + n, rtol=5e-4)


@pytest.mark.parametrize('K,M,N', [(8192, 256, 64), (8192, 256, 52), (8001, 52, 64), (5, 7, 3), (1, 16, 16), (300, 159, 100),
                                   (70000, 96, 32), (100, 33, 96), (15360, 256, 128)])
def test_atb_matches_fp64(K, M, N):
    """p2c_atb: C = A^T B and the column sums of A (weight + bias gradient of a dense layer), overwrite and accumulate,
    contiguous and row-strided operands, vs fp64."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    torch.manual_seed(K + M + N)
    a, b = torch.randn(K, M, device=d), torch.randn(K, N, device=d)
    c, cb = ops.atb(a, b, bias=True)
    ref, refb = a.double().t() @ b.double(), a.double().sum(0)
    close(c, ref, 'A^T B', rtol=2e-6 * max(1.0, K ** 0.5))
    close(cb, refb, 'column sums', rtol=2e-6 * max(1.0, K ** 0.5))
    c2 = c.clone()
    ops.atb(a, b, out=c2, accumulate=True)
    close(c2, 2 * ref, 'accumulate', rtol=2e-6 * max(1.0, K ** 0.5))
    c4, b4 = c.clone(), cb.clone()
    ops.atb(a, b, bias=True, out=c4, bias_out=b4, accumulate=True)       # both outputs given: both accumulate
    close(c4, 2 * ref, 'accumulate C', rtol=2e-6 * max(1.0, K ** 0.5)), close(b4, 2 * refb, 'accumulate bias', rtol=2e-6 * max(1.0, K ** 0.5))
    c5, b5 = ops.atb(a, b, bias=True, out=c.clone(), accumulate=True)    # fresh bias vector: overwritten, not accumulated
    close(b5, refb, 'fresh bias', rtol=2e-6 * max(1.0, K ** 0.5))
    wide_a, wide_b = torch.randn(K, M + 5, device=d), torch.randn(K, N + 9, device=d)
    c3, _ = ops.atb(wide_a[:, 2:2 + M], wide_b[:, 4:4 + N])
    close(c3, wide_a[:, 2:2 + M].double().t() @ wide_b[:, 4:4 + N].double(), 'strided', rtol=2e-6 * max(1.0, K ** 0.5))
    assert torch.equal(ops.atb(a, b)[0], ops.atb(a, b)[0])              # fixed summation order


@pytest.mark.parametrize('T,B,I,H', [(1, 3, 5, 16), (16, 37, 52, 64), (9, 130, 20, 32)])
def test_layer_with_the_default_zero_state(T, B, I, H):
    """h0 = c0 = None (nn.LSTM's default): the kernel reads no initial state, the W_hh gradient has no t = 0 term and no
    state gradients are produced -- vs torch.nn.LSTM called without a state, fp64."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    torch.manual_seed(T + B + H)
    ref = torch.nn.LSTM(I, H).double()
    x = torch.randn(T, B, I, dtype=torch.float64)
    up, uh, uc = torch.randn(T, B, H, dtype=torch.float64), torch.randn(B, H, dtype=torch.float64), torch.randn(B, H, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    out_r, (hT_r, cT_r) = ref(xr)
    ((out_r * up).sum() + (hT_r[0] * uh).sum() + (cT_r[0] * uc).sum()).backward()
    p = {n: v.detach().float().to(d).requires_grad_(True) for n, v in ref.named_parameters()}
    xd = x.float().to(d).requires_grad_(True)
    out, hT, cT = ops.lstm_layer(xd, None, None, p['weight_ih_l0'], p['weight_hh_l0'], p['bias_ih_l0'], p['bias_hh_l0'])
    ((out * up.float().to(d)).sum() + (hT * uh.float().to(d)).sum() + (cT * uc.float().to(d)).sum()).backward()
    close(out, out_r, 'out'), close(hT, hT_r[0], 'hT'), close(cT, cT_r[0], 'cT')
    close(xd.grad, xr.grad, 'grad x')
    for n, v in ref.named_parameters():
        close(p[n].grad, v.grad, 'grad ' + n)


def test_weight_gradients_added_straight_into_the_flat_buffer_equal_autograd_accumulation():
    """Flat trainer (ops.grad_sinks: K12 adds dW into the views of the flat gradient buffer, returns no gradient to autograd)
    vs the same model with per-parameter gradients through autograd: every gradient of one Seq2SeqEmbeddings train step."""
    from pedestrians_video_2_carla_amd import ops
    from pedestrians_video_2_carla_amd.data.carla.carla_recorded_synthetic import SyntheticCarlaRecordedDataModule
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.autoencoder import LitAutoencoderFlow
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2SeqEmbeddings
    from pedestrians_video_2_carla_amd.trainer import Trainer, seed_everything
    d = dev()
    grads = {}
    for flat in (True, False):
        seed_everything(12)
        dm = SyntheticCarlaRecordedDataModule(clip_length=16, batch_size=80)      # 1 280 rows: the K12 path (>= 1 024)
        model = Seq2SeqEmbeddings(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_2d,
                                  p_dropout=0.0)
        flow = LitAutoencoderFlow(movements_model=model, loss_modes=['loc_2d'], transform='hips_neck_bbox')
        trainer = Trainer(device=d, use_graph=False, flatten=flat).setup(flow, dm)
        assert trainer._grad_sinks == flat and not ops.GRAD_SINKS      # a per-trainer context, not a process global
        batch = dm.generate_batch(d)
        flow.train()
        with ops.grad_sinks(trainer._grad_sinks):                        # what Trainer._forward_backward opens
            flow.on_train_batch_start(batch, 0)
            flow.training_step(batch, 0)['loss'].backward()
        grads[flat] = {n: p.grad.detach().clone() for n, p in flow.named_parameters()}
    assert not ops.GRAD_SINKS
    for n in grads[True]:
        close(grads[True][n], grads[False][n], 'grad ' + n, rtol=2e-5)


def test_cfg3_batch_size_parity_with_the_cpu_twin():
    """BASELINE.json configs[2] at its own batch size: autoencoder flow, Seq2SeqEmbeddings(pose_2d), B = 512, T = 16 -- loss and
    every parameter gradient of one training step on the GPU (folded embeddings, K7b recurrences, K7c decoder loop, K12 weight
    gradients, K3 loss) vs the same flow in fp64 on the CPU; tolerance max(1e-4, 2 x what fp32 on the CPU loses)."""
    import copy
    from pedestrians_video_2_carla_amd.data.carla.carla_recorded_synthetic import SyntheticCarlaRecordedDataModule
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.autoencoder import LitAutoencoderFlow
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2SeqEmbeddings
    from pedestrians_video_2_carla_amd.trainer import seed_everything
    from oracle import pose_head as O
    d = dev()
    seed_everything(22742)
    dm = SyntheticCarlaRecordedDataModule(clip_length=16, batch_size=512, missing_joint_probabilities=0.1)
    model = Seq2SeqEmbeddings(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_2d,
                              p_dropout=0.0)
    twins = {torch.float64: copy.deepcopy(model).double(), torch.float32: copy.deepcopy(model)}
    flow = LitAutoencoderFlow(movements_model=model, loss_modes=['loc_2d'], transform='hips_neck_bbox').to(d).train()
    flow.attach_datamodule(dm) if hasattr(flow, 'attach_datamodule') else None
    batch = dm.generate_batch(d)
    frames, targets, meta = batch
    flow.on_train_batch_start(batch, 0)
    out = flow.training_step(batch, 0)
    out['loss'].backward()
    ref = {}
    for dt, twin in twins.items():
        twin.train()
        pred = twin(frames.to('cpu', dt))
        loss = O.loss_loc_2d(pred, targets['projection_2d_transformed'].to('cpu', dt))[0]
        loss.backward()
        ref[dt] = (loss.detach(), [p.grad for p in twin.parameters()])
    l64, g64 = ref[torch.float64]
    l32, g32 = ref[torch.float32]
    close(out['loss'], l64, 'loss', rtol=max(1e-4, 2 * abs(float(l32) - float(l64)) / abs(float(l64))))
    for (n, p), q, q32 in zip(flow.movements_model.named_parameters(), g64, g32):
        ref_err = (q32.double() - q).abs().max().item() / (q.abs().max().item() + 1e-30)
        close(p.grad, q, 'grad ' + n, rtol=max(1e-4, 2 * ref_err))
