"""Grouped per-joint embeddings (K7a, p2c_embed_fwd/_bwd through the C ABI) against the reference formula in fp64:
y[t', b, j, :] = W_j x[b, t, j, :] + b_j (seq2seq_embeddings.py:53-78), sequence-first, optional time reversal.
Tolerance 1e-4 relative (fp32); the gradient reduction is deterministic (bitwise equal across runs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def close(a, b, what, rtol=RTOL):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err, scale = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= rtol * scale + 1e-30, f'{what}: {err:.3e} vs scale {scale:.3e}'


def reference(x, ws, bs, flip):
    W, b = torch.stack(ws).double(), torch.stack(bs).double()
    emb = torch.einsum('btjc,jec->tbje', x.double(), W) + b
    return emb.flip(0) if flip else emb


@pytest.mark.parametrize('B,T,J,C,E', [(1, 1, 26, 2, 64), (5, 16, 26, 2, 64), (33, 7, 25, 3, 32), (300, 16, 26, 2, 64),
                                       (3, 4, 18, 4, 128)])
@pytest.mark.parametrize('flip', [False, True])
@pytest.mark.parametrize('layout', ['separate', 'flat'])
def test_forward_backward_match_reference(B, T, J, C, E, flip, layout):
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(B * 131 + T)
    x = torch.randn(B, T, J, C, generator=g)
    if layout == 'flat':        # views of one buffer, interleaved w_0, b_0, w_1, b_1, ... as in the flat trainer
        flat = (torch.randn(J * (E * C + E), generator=g) * 0.3).to(d)
        ws = [flat[j * (E * C + E): j * (E * C + E) + E * C].view(E, C).requires_grad_(True) for j in range(J)]
        bs = [flat[j * (E * C + E) + E * C: (j + 1) * (E * C + E)].requires_grad_(True) for j in range(J)]
    else:
        ws = [(torch.randn(E, C, generator=g) * 0.3).to(d).requires_grad_(True) for _ in range(J)]
        bs = [(torch.randn(E, generator=g) * 0.3).to(d).requires_grad_(True) for _ in range(J)]
    up = torch.randn(T, B, J, E, generator=g)
    y = ops.joint_embeddings(x.to(d), ws, bs, flip=flip)
    (y * up.to(d)).sum().backward()
    wr = [w.detach().cpu().double().requires_grad_(True) for w in ws]
    br = [b.detach().cpu().double().requires_grad_(True) for b in bs]
    yr = reference(x, wr, br, flip)
    (yr * up.double()).sum().backward()
    close(y, yr, 'y')
    close(torch.stack([w.grad for w in ws]), torch.stack([w.grad for w in wr]), 'grad W')
    close(torch.stack([b.grad for b in bs]), torch.stack([b.grad for b in br]), 'grad b')


def test_gradient_sink_and_determinism():
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    B, T, J, C, E = 64, 16, 26, 2, 64
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, T, J, C, generator=g).to(d)
    flat = (torch.randn(J * (E * C + E), generator=g) * 0.3).to(d)
    gflat = torch.full_like(flat, 7.0)                      # the sink is overwritten, not accumulated into
    sl = lambda buf, j: (buf[j * (E * C + E): j * (E * C + E) + E * C].view(E, C), buf[j * (E * C + E) + E * C: (j + 1) * (E * C + E)])
    ws = [sl(flat, j)[0].requires_grad_(True) for j in range(J)]
    bs = [sl(flat, j)[1].requires_grad_(True) for j in range(J)]
    sinks = [t for j in range(J) for t in sl(gflat, j)]
    up = torch.randn(T, B, J, E, generator=g).to(d)
    (ops.joint_embeddings(x, ws, bs, sinks=sinks) * up).sum().backward()
    assert all(w.grad is None for w in ws)                   # nothing returned to autograd
    first = gflat.clone()
    ws2 = [w.detach().clone().requires_grad_(True) for w in ws]
    bs2 = [b.detach().clone().requires_grad_(True) for b in bs]
    (ops.joint_embeddings(x, ws2, bs2) * up).sum().backward()
    want = torch.cat([torch.cat([w.grad.reshape(-1), b.grad.reshape(-1)]) for w, b in zip(ws2, bs2)])
    assert torch.equal(first, want), 'sink path and autograd path run the same deterministic reduction'


def test_errors():
    from pedestrians_video_2_carla_amd import ops, _lib
    d = dev()
    ws = [torch.zeros(64, 2, device=d) for _ in range(26)]
    bs = [torch.zeros(64, device=d) for _ in range(26)]
    with pytest.raises(_lib.P2CError):                       # host tensor: no CPU fallback
        ops.joint_embeddings(torch.zeros(2, 4, 26, 2), ws, bs)
    with pytest.raises(RuntimeError):
        ops.joint_embeddings(torch.zeros(2, 4, 25, 2, device=d), ws, bs)
