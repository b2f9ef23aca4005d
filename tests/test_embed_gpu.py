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


@pytest.mark.gpu
@pytest.mark.parametrize('G,J,E,C,acc', [(256, 26, 64, 2, False), (7, 5, 20, 3, True), (33, 3, 130, 4, False), (1, 1, 4, 1, True)])
def test_fold_kernels_match_fp64(G, J, E, C, acc):
    """K7a' (p2c_fold_fwd / p2c_fold_bwd) on cfg3's shape and on odd ones (gate rows not a multiple of the 16 row groups, more
    than 64 channels per joint, C = 1..4), strided parameter blocks, against the composition written out in fp64: w_eff, b_eff,
    and from given output gradients d W_ih (written or accumulated), d W_j / d b_j (accumulated) and the two bias gradients."""
    import ctypes
    import torch
    from pedestrians_video_2_carla_amd import _lib, ops
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(G * 7 + J)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    w_ih, W, b, b_ih, b_hh = rnd(G, J * E), rnd(J, E, C), rnd(J, E), rnd(G), rnd(G)
    g_eff, g_b = rnd(G, J * C), rnd(G)
    w3 = w_ih.view(G, J, E)
    w_eff_ref = torch.einsum('gje,jec->gjc', w3, W).reshape(G, J * C)
    b_eff_ref = torch.einsum('gje,je->g', w3, b) + b_ih + b_hh
    g3 = g_eff.view(G, J, C)
    g_w_ref = (torch.einsum('gjc,jec->gje', g3, W) + g_b.view(G, 1, 1) * b.unsqueeze(0)).reshape(G, J * E)
    gW_ref, gb_ref = torch.einsum('gjc,gje->jec', g3, w3), torch.einsum('g,gje->je', g_b, w3)

    f = lambda t: t.float().to(d).contiguous()
    pad = 3                                                     # parameter blocks with a stride, like views of a flat buffer
    Wb, bb = torch.zeros(J, E * C + pad, device=d), torch.zeros(J, E + pad, device=d)
    Wb[:, :E * C], bb[:, :E] = f(W).view(J, -1), f(b)
    gWb, gbb = torch.ones_like(Wb), torch.ones_like(bb)
    w_ih_d, b_ih_d, b_hh_d = f(w_ih), f(b_ih), f(b_hh)
    w_eff, b_eff = torch.empty(G, J * C, device=d), torch.empty(G, device=d)
    lib, st = _lib.lib(), ops._stream()
    _lib.check(lib.p2c_fold_fwd(w_ih_d.data_ptr(), Wb.data_ptr(), bb.data_ptr(), Wb.stride(0), bb.stride(0), b_ih_d.data_ptr(),
                                b_hh_d.data_ptr(), w_eff.data_ptr(), b_eff.data_ptr(), G, J, E, C, st), 'p2c_fold_fwd')
    g_w = torch.ones(G, J * E, device=d) if acc else torch.empty(G, J * E, device=d)
    g_bi, g_bh = torch.ones(G, device=d), torch.ones(G, device=d)
    g_eff_d, g_b_d = f(g_eff), f(g_b)                            # (kept alive: the library sees raw pointers)
    _lib.check(lib.p2c_fold_bwd(w_ih_d.data_ptr(), Wb.data_ptr(), bb.data_ptr(), Wb.stride(0), bb.stride(0), g_eff_d.data_ptr(),
                                g_b_d.data_ptr(), g_w.data_ptr(), int(acc), gWb.data_ptr(), gbb.data_ptr(), g_bi.data_ptr(),
                                g_bh.data_ptr(), G, J, E, C, st), 'p2c_fold_bwd')

    def close(a, ref, what, rtol=2e-5):
        a, ref = a.double().cpu(), ref
        err, sc = (a - ref).abs().max().item(), max(ref.abs().max().item(), 1e-6)
        assert err <= rtol * sc * max(1.0, G ** 0.5), f'{what}: {err:.3e} vs scale {sc:.3e}'
    close(w_eff, w_eff_ref, 'w_eff'), close(b_eff, b_eff_ref, 'b_eff')
    close(g_w, g_w_ref + (1 if acc else 0), 'd w_ih')
    close(gWb[:, :E * C].reshape(J, E, C), gW_ref + 1, 'd W_j'), close(gbb[:, :E], gb_ref + 1, 'd b_j')
    assert (gWb[:, E * C:] == 1).all() and (gbb[:, E:] == 1).all()          # the padding between the blocks is untouched
    close(g_bi, g_b + 1, 'd b_ih'), close(g_bh, g_b + 1, 'd b_hh')


@pytest.mark.gpu
def test_group_copy_matches_per_tensor_copies():
    """p2c_copy_group through ops.GroupCopy: 30 tensors (two launches of <= 24), sizes from 0 to 3 MB, odd byte counts and
    int32 / uint8 payloads, a non-contiguous source -- every destination equals its source, nothing else is touched."""
    import torch
    from pedestrians_video_2_carla_amd import ops
    d = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(1)
    shapes = [(0,), (1,), (3,), (5, 7), (256, 16, 26, 2), (256, 16, 26, 3, 3), (1025,), (17, 3)] * 3 + [(33,), (2, 2), (9,), (4, 4, 4), (1,), (7,)]
    dst, src = [], []
    for i, shp in enumerate(shapes):
        if i % 7 == 3:
            s = torch.randint(0, 100, shp, generator=g, dtype=torch.int32).to(d)
        elif i % 7 == 5:
            s = torch.randint(0, 255, shp, generator=g, dtype=torch.uint8).to(d)
        else:
            s = torch.randn(*shp, generator=g).to(d)
        src.append(s)
        dst.append(torch.full_like(s, 7))
    src[4] = src[4].transpose(1, 2).contiguous().transpose(1, 2)          # same values, non-contiguous view
    guard = torch.full((1024,), 3.0, device=d)
    copy = ops.GroupCopy(dst)
    copy(src)
    torch.cuda.synchronize()
    for a, b in zip(dst, src):
        assert torch.equal(a, b)
    assert (guard == 3).all()
