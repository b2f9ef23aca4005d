"""K12 at the spatial-block shapes of cfg5 (546 624 rows):  python tools/atb_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import ops
d = torch.device('cuda:0')
K = 546624
for M, N in ((96, 32), (32, 32), (64, 32), (32, 64)):
    a, b = torch.randn(K, M, device=d), torch.randn(K, N, device=d)
    for _ in range(3):
        ops.atb(a, b, bias=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.atb(a, b, bias=True)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) * 100
    print(f'K={K} M={M} N={N}: {t:.1f} us, {(M + N) * K * 4 / t / 1e6:.2f} TB/s')
print('K12 with the per-sample factor on A (rows_per_scale = 26)')
for M, N in ((96, 32), (32, 32), (64, 32), (32, 64)):
    a, b = torch.randn(K, M, device=d), torch.randn(K, N, device=d)
    f = torch.rand(K // 26, device=d)
    for _ in range(3):
        ops.atb(a, b, bias=True, a_scale=f, rows_per_scale=26)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.atb(a, b, bias=True, a_scale=f, rows_per_scale=26)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) * 100
    print(f'K={K} M={M} N={N}: {t:.1f} us, {(M + N) * K * 4 / t / 1e6:.2f} TB/s')
print('K16 TN form')
for M, N in ((96, 32), (32, 32), (64, 32), (32, 64)):
    a, b = torch.randn(K, M, device=d), torch.randn(K, N, device=d)
    for _ in range(3):
        ops.gemm_tn(a, b, bias=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_tn(a, b, bias=True)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) * 100
    print(f'K={K} M={M} N={N}: {t:.1f} us, {(M + N) * K * 4 / t / 1e6:.2f} TB/s')
