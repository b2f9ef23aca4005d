"""Developer tool: phase timeline of the fused-MLP kernels (needs a build with EXTRA=-DP2C_MLP_TRACE)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import _lib, ops

d = torch.device('cuda:0')
dims = [52, 26, 13, 6, 39, 78, 156]
Ws = [(torch.randn(o, i, device=d) * 0.1).requires_grad_(True) for i, o in zip(dims[:-1], dims[1:])]
bs = [(torch.randn(o, device=d) * 0.1).requires_grad_(True) for o in dims[1:]]
lib = _lib.lib()
lib.p2c_debug_mlp_trace.argtypes = [ctypes.c_void_p]
for N in [int(a) for a in sys.argv[1:]] or [16, 4096]:
    x = torch.randn(N, dims[0], device=d)
    gy = torch.randn(N, dims[-1], device=d)
    sinks = [torch.zeros_like(t) for pair in zip(Ws, bs) for t in pair]
    for _ in range(5):
        ops.FusedMLPFunction.apply(x, len(Ws), sinks, None, False, None, 0, *Ws, *bs).backward(gy)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 80)()
    assert lib.p2c_debug_mlp_trace(buf) == 0
    for k, name in ((0, 'fwd'), (1, 'bwd')):
        t = list(buf[40 * k:40 * k + 40])
        cyc, wall = t[39] - t[0], (t[37] - t[38]) * 10.0       # wall clock: 100 MHz -> ns
        print(f'N={N} {name}: {cyc} cycles, {wall:.0f} ns, {cyc / max(wall, 1):.2f} GHz')
        order = sorted((v, i) for i, v in enumerate(t[:37]) if v >= t[0] and i > 0) + [(t[39], 39)]
        prev = t[0]
        for v, i in order:
            print(f'   [{i:2d}] +{v - prev:6d}')
            prev = v
