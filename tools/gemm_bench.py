"""K16 (ops.gemm, fp32 MFMA) against the library GEMM at the shapes of cfg5 (PoseFormer) and cfg3 (Seq2SeqEmbeddings):
   python tools/gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import ops


def t_us(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


d = torch.device('cuda:0')
for M, N, K, tb in ((21024, 2496, 832, True), (21024, 832, 832, True), (21024, 1664, 832, True), (21024, 832, 1664, True),
                    (21024, 832, 2496, False), (21024, 832, 1664, False), (546624, 96, 32, True), (546624, 32, 32, True),
                    (546624, 64, 32, True), (546624, 32, 64, True), (546624, 32, 96, False), (8192, 256, 52, True),
                    (8192, 256, 64, True), (8192, 64, 256, False)):
    a = torch.randn(M, K, device=d)
    b = torch.randn(N, K, device=d) if tb else torch.randn(K, N, device=d)
    out = torch.empty(M, N, device=d)
    own = t_us(lambda: ops.gemm(a, b, tb, out=out))
    lib = t_us(lambda: torch.mm(a, b.t() if tb else b, out=out))
    fl = 2.0 * M * N * K
    print(f'M={M:7d} N={N:5d} K={K:5d} {"NT" if tb else "NN"}  K16 {own:9.1f} us {fl / own / 1e6:7.1f} TF   library {lib:9.1f} us {fl / lib / 1e6:7.1f} TF', flush=True)
