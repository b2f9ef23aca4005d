"""GPU: K16 (csrc/p2c_gemm.hip) -- dense layers on fp32 MFMA with the fused epilogue -- through the C ABI, against fp64 formulas.

Reference binding: the nn.Linear layers of the model plugins (PoseTransformer qkv / proj / fc1 / fc2 behind
modules/movements/pose_former/pose_former.py:62-76; the LSTM input projections of movements/seq2seq/seq2seq.py:36-38,72-73).
fp32 MFMA is an fmaf chain in k order: errors are fp32 rounding of a K-term dot product, bounds below are 2e-6 * sum|a||b| scale."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def gelu64(z):
    return 0.5 * z * (1 + torch.erf(z / math.sqrt(2.0)))


@pytest.mark.parametrize('M,N,K', [(21024, 2496, 832), (4099, 96, 32), (1000, 32, 96), (777, 52, 64), (130, 64, 52), (8192, 256, 52),
                                   (5, 3, 7), (1, 1, 1), (257, 129, 33), (300, 70, 1664),
                                   # K % 32 == 0 and aligned (the buffer-load forms) with ragged M and N tiles
                                   (300, 100, 64), (129, 36, 96), (1000, 68, 32), (131, 260, 128)])
@pytest.mark.parametrize('trans_b', [True, False])
def test_gemm_plain_matches_fp64(M, N, K, trans_b):
    from pedestrians_video_2_carla_amd import ops
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev())
    b = torch.randn(N, K, device=dev()) if trans_b else torch.randn(K, N, device=dev())
    c = ops.gemm(a, b, trans_b)
    want = a.double() @ (b.double().t() if trans_b else b.double())
    assert rel(c, want) < 2e-6 * math.sqrt(K) + 1e-7


@pytest.mark.parametrize('K,M,N', [(21024, 2496, 832), (21024, 832, 1664), (4000, 130, 200), (31, 129, 140), (1, 5, 3), (9000, 64, 33),
                                   (2050, 257, 129), (4096, 132, 100), (640, 36, 260)])    # (the last two: buffer-load forms, ragged tiles)
def test_gemm_tn_matches_fp64_is_reproducible_and_accumulates(K, M, N):
    """dW = dy^T x for wide layers: split-K slabs added in a fixed order -- same bits on every call; accumulate adds to the sink;
    strided operands (column slices) take the dword loads."""
    from pedestrians_video_2_carla_amd import ops
    torch.manual_seed(K + M)
    a, b = torch.randn(K, M, device=dev()), torch.randn(K, N, device=dev())
    c = ops.gemm_tn(a, b)
    want = a.double().t() @ b.double()
    assert rel(c, want) < 2e-6 * math.sqrt(K) + 1e-7
    assert torch.equal(c, ops.gemm_tn(a, b))
    sink = torch.randn(M, N, device=dev())
    before = sink.clone()
    ops.gemm_tn(a, b, out=sink, accumulate=True)
    assert rel(sink, before.double() + want) < 2e-6 * math.sqrt(K) + 1e-7
    wide_a, wide_b = torch.randn(K, M + 3, device=dev()), torch.randn(K, N + 5, device=dev())
    c2 = ops.gemm_tn(wide_a[:, 1:M + 1], wide_b[:, 2:N + 2])
    assert rel(c2, wide_a[:, 1:M + 1].double().t() @ wide_b[:, 2:N + 2].double()) < 2e-6 * math.sqrt(K) + 1e-7


@pytest.mark.parametrize('K,M,N,per', [(9 * 500, 832, 256, 9), (26 * 300, 140, 200, 26), (26 * 300, 32, 96, 26), (64, 5, 3, 1),
                                       (26 * 3000, 96, 32, 26)])       # (the last: a skinny dW over many rows -> ~150 K slices)
def test_weight_gradient_with_row_factor_and_bias_in_one_pass(K, M, N, per):
    """dW = (dy * f)^T x and db = column sums of dy * f, f per sample (K12 below 128 features, the K16 TN form above), written
    and accumulated; same bits on every call."""
    from pedestrians_video_2_carla_amd import ops
    torch.manual_seed(K + N)
    a, b = torch.randn(K, M, device=dev()), torch.randn(K, N, device=dev())
    f = (torch.rand(K // per, device=dev()) > 0.3).float() / 0.7
    a64 = a.double() * f.double().repeat_interleave(per).view(-1, 1)
    for fn in (ops.gemm_tn, ops.atb):
        c, db = fn(a, b, bias=True, a_scale=f, rows_per_scale=per)
        assert rel(c, a64.t() @ b.double()) < 2e-6 * math.sqrt(K) + 1e-7
        assert rel(db, a64.sum(0)) < 2e-6 * math.sqrt(K) + 1e-7
        c2, db2 = fn(a, b, bias=True, a_scale=f, rows_per_scale=per)
        assert torch.equal(c, c2) and torch.equal(db, db2)
        sink, bsink = torch.randn(M, N, device=dev()), torch.randn(M, device=dev())
        w0, b0 = sink.clone(), bsink.clone()
        fn(a, b, bias=True, out=sink, bias_out=bsink, accumulate=True, a_scale=f, rows_per_scale=per)
        assert rel(sink, w0.double() + a64.t() @ b.double()) < 2e-6 * math.sqrt(K) + 1e-7
        assert rel(bsink, b0.double() + a64.sum(0)) < 2e-6 * math.sqrt(K) + 1e-7


def test_gemm_epilogue_and_strided_operands():
    """bias, GELU with the stored pre-activation, gelu' of a stored tensor, per-sample factor, residual (also aliased with the
    output), operands that are column slices of wider tensors (leading dimension > width; not 16-byte aligned -> dword loads)."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    torch.manual_seed(3)
    M, N, K, rows_per = 9 * 211, 200, 72, 9
    A = torch.randn(M, K + 5, device=d)[:, 1:K + 1]                 # lda = K + 5, base off by 4 bytes
    W = torch.randn(N, K + 4, device=d)[:, :K]
    bias, res = torch.randn(N, device=d), torch.randn(M, N, device=d)
    scale = (torch.rand(M // rows_per, device=d) > 0.3).float() / 0.7
    z = torch.empty(M, N, device=d)
    y = ops.gemm(A, W, True, bias=bias, act=1, aux_out=z, row_scale=scale, rows_per_scale=rows_per, residual=res)
    z64 = A.double() @ W.double().t() + bias.double()
    want = gelu64(z64) * scale.double().repeat_interleave(rows_per).view(-1, 1) + res.double()
    assert rel(z, z64) < 2e-5 and rel(y, want) < 2e-5
    # backward-side epilogue: (gy W) * gelu'(z) * scale, gy a strided view, output written over the residual
    gy = torch.randn(M, N + 8, device=d)[:, 4:N + 4]
    dz_res = torch.randn(M, K, device=d)
    zz = torch.randn(M, K, device=d)
    out = dz_res.clone()
    ops.gemm(gy, W.contiguous(), False, act=2, aux=zz, row_scale=scale, rows_per_scale=rows_per, residual=out, out=out)
    z6 = zz.double()
    gelu_grad = 0.5 * (1 + torch.erf(z6 / math.sqrt(2.0))) + z6 * torch.exp(-0.5 * z6 * z6) / math.sqrt(2 * math.pi)
    want = (gy.double() @ W.double()) * gelu_grad * scale.double().repeat_interleave(rows_per).view(-1, 1) + dz_res.double()
    assert rel(out, want) < 2e-5


@pytest.mark.parametrize('rows,din,dhid,per', [(2336 * 9, 832, 1664, 9), (26 * 500, 32, 64, 26), (333, 52, 40, 1)])
def test_fused_dense_and_mlp_gradients_match_autograd_fp64(rows, din, dhid, per):
    """ops.dense / ops.mlp_gelu (K16 + K12) with the stochastic-depth factor and the residual: output and every gradient against
    torch autograd of the written-out formula in fp64."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    torch.manual_seed(rows)
    x = torch.randn(rows, din, device=d) * 0.5
    res = torch.randn(rows, din, device=d)
    w1, b1 = torch.randn(dhid, din, device=d) / math.sqrt(din), torch.randn(dhid, device=d) * 0.1
    w2, b2 = torch.randn(din, dhid, device=d) / math.sqrt(dhid), torch.randn(din, device=d) * 0.1
    scale = (torch.rand(rows // per, device=d) > 0.25).float() / 0.75
    up = torch.randn(rows, din, device=d)
    leaves = [t.clone().requires_grad_(True) for t in (x, res, w1, b1, w2, b2)]
    y = ops.mlp_gelu(leaves[0], leaves[2], leaves[3], leaves[4], leaves[5], scale, per, leaves[1])
    (y * up).sum().backward()
    l64 = [t.double().clone().requires_grad_(True) for t in (x, res, w1, b1, w2, b2)]
    s64 = scale.double().repeat_interleave(per).view(-1, 1)
    y64 = (gelu64(l64[0] @ l64[2].t() + l64[3]) @ l64[4].t() + l64[5]) * s64 + l64[1]
    (y64 * up.double()).sum().backward()
    assert rel(y, y64) < 2e-5
    for name, a, b in zip(('x', 'residual', 'w1', 'b1', 'w2', 'b2'), leaves, l64):
        assert rel(a.grad, b.grad) < 5e-5, name
    # the single layer: y = (x W^T + b) * scale + residual
    leaves = [t.clone().requires_grad_(True) for t in (x, res[:, :1].expand(rows, dhid).contiguous(), w1, b1)]
    y = ops.dense(leaves[0], leaves[2], leaves[3], scale, per, leaves[1])
    upd = torch.randn(rows, dhid, device=d)
    (y * upd).sum().backward()
    l64 = [t.detach().double().clone().requires_grad_(True) for t in leaves]
    y64 = (l64[0] @ l64[2].t() + l64[3]) * s64 + l64[1]
    (y64 * upd.double()).sum().backward()
    assert rel(y, y64) < 2e-5
    for name, a, b in zip(('x', 'residual', 'w', 'b'), leaves, l64):
        assert rel(a.grad, b.grad) < 5e-5, name


def test_gemm_rejects_bad_arguments():
    from pedestrians_video_2_carla_amd import _lib, ops
    d = dev()
    a, b = torch.zeros(4, 8, device=d), torch.zeros(5, 9, device=d)
    with pytest.raises(RuntimeError):
        ops.gemm(a, b, True)                                   # inner dimensions differ
    with pytest.raises(_lib.P2CError):
        ops.gemm(torch.zeros(4, 8), torch.zeros(5, 8), True)   # host tensors: no CPU fallback
    with pytest.raises(_lib.P2CError):
        ops.gemm(a, torch.zeros(5, 8, device=d), True, act=2)  # gelu' needs the stored pre-activation
