"""K16's TN form (weight gradients of wide layers) at cfg5's shapes: the slice count the host picks against a sweep of forced
counts (P2C_GEMM_TN_SLICES) and the library.   python tools/tn_bench.py [sweep]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import _lib
from pedestrians_video_2_carla_amd import ops
d = torch.device('cuda:0')


def t_us(f, n=10):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K, M, N in [(21024, 2496, 832), (21024, 832, 832), (21024, 1664, 832), (21024, 832, 1664)]:
    a, b = torch.randn(K, M, device=d), torch.randn(K, N, device=d)
    os.environ.pop('P2C_GEMM_TN_SLICES', None)
    _lib.lib().p2c_gemm_reload_env()
    own, lib = t_us(lambda: ops.gemm_tn(a, b)), t_us(lambda: torch.mm(a.t(), b))
    line = f'K={K} M={M} N={N}: picked {own:7.1f} us {2 * K * M * N / own / 1e6:6.1f} TF | library {lib:7.1f} us {2 * K * M * N / lib / 1e6:6.1f} TF'
    if len(sys.argv) > 1:
        res = []
        for s in range(1, 41):
            os.environ['P2C_GEMM_TN_SLICES'] = str(s)
            _lib.lib().p2c_gemm_reload_env()
            res.append((t_us(lambda: ops.gemm_tn(a, b), 5), s))
        os.environ.pop('P2C_GEMM_TN_SLICES', None)
        _lib.lib().p2c_gemm_reload_env()
        res.sort()
        line += ' | sweep best: ' + ', '.join(f's={s}: {t:.0f}' for t, s in res[:5]) + ' | worst: ' + ', '.join(f's={s}: {t:.0f}' for t, s in res[-2:])
    print(line, flush=True)
