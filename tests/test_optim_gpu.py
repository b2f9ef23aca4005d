"""FlatAdamW (p2c_adamw_step) against torch.optim.AdamW / Adam on the same gradients. fp32; tolerance 5e-6 relative after
25 steps (same formula, bias corrections in double as in ATen; only the fp32 rounding order inside the moment updates
differs) -- well inside the 1e-4 the north star allows."""
TOL = 5e-6
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('n', [1, 7, 17530, 70001])
@pytest.mark.parametrize('decoupled', [True, False])
def test_matches_torch(n, decoupled):
    from pedestrians_video_2_carla_amd.parallel.optim import FlatAdamW
    d = dev()
    g = torch.Generator(device=d).manual_seed(n)
    p0 = torch.randn(n, device=d, generator=g)
    ours = torch.nn.Parameter(p0.clone())
    ref = torch.nn.Parameter(p0.clone())
    kw = dict(lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    o = FlatAdamW([ours], decoupled=decoupled, zero_grad_in_step=True, **kw)
    r = (torch.optim.AdamW if decoupled else torch.optim.Adam)([ref], **kw)
    ours.grad = torch.zeros_like(ours)
    for step in range(25):
        grad = torch.randn(n, device=d, generator=g) * (1.0 + step)
        ours.grad.add_(grad)                       # accumulate into the zeroed buffer, as autograd does
        ref.grad = grad.clone()
        o.step()
        r.step()
        assert float(ours.grad.abs().max()) == 0.0  # left zeroed for the next step
    assert rel(ours.data, ref.data) < TOL
    st, rt = o.state[ours], r.state[ref]
    assert float(st['step']) == 25.0 == float(rt['step'])
    assert rel(st['exp_avg'], rt['exp_avg']) < TOL and rel(st['exp_avg_sq'], rt['exp_avg_sq']) < TOL


def test_grad_scale_lr_change_and_graph_replay():
    from pedestrians_video_2_carla_amd.parallel.optim import FlatAdamW
    d = dev()
    g = torch.Generator(device=d).manual_seed(3)
    n = 4099
    p0 = torch.randn(n, device=d, generator=g)
    ours, ref = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    o = FlatAdamW([ours], lr=1e-2, weight_decay=0.01, zero_grad_in_step=False)
    o.grad_scale = 0.25                                  # data-parallel averaging over 4 ranks folded into the step
    r = torch.optim.AdamW([ref], lr=1e-2, weight_decay=0.01)
    grads = [torch.randn(n, device=d, generator=g) for _ in range(6)]
    ours.grad = torch.zeros_like(ours)
    static = ours.grad
    o.sync_hyper()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            o.step()
    torch.cuda.current_stream().wait_stream(side)
    for i, gr in enumerate(grads):
        if i == 3:                                       # scheduler step between replays
            o.param_groups[0]['lr'] = r.param_groups[0]['lr'] = 2e-3
            o.sync_hyper()
        static.copy_(gr * 4.0)
        graph.replay()
        ref.grad = gr.clone()
        r.step()
    torch.cuda.synchronize()
    assert float(o.state[ours]['step']) == 6.0
    assert rel(ours.data, ref.data) < TOL


def test_state_dict_round_trip_and_no_cpu_fallback():
    from pedestrians_video_2_carla_amd import _lib
    from pedestrians_video_2_carla_amd.parallel.optim import FlatAdamW
    d = dev()
    p = torch.nn.Parameter(torch.randn(64, device=d))
    o = FlatAdamW([p])
    p.grad = torch.randn(64, device=d)
    o.step()
    sd = o.state_dict()
    assert set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
    q = torch.nn.Parameter(p.data.clone())
    o2 = FlatAdamW([q])
    o2.load_state_dict(sd)
    assert float(o2.state[q]['step']) == 1.0
    with pytest.raises(_lib.P2CError):
        FlatAdamW([torch.nn.Parameter(torch.randn(8))])
