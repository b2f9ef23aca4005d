"""Which BLAS backend for the Seq2Seq weight-gradient shapes? (dW = dgates^T x over T*B rows)"""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = torch.device('cuda:0')
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5): g.replay()
        e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 5)
shapes = [(256, 8192, 64), (256, 8192, 52), (256, 8192, 1664), (52, 8192, 64), (8192, 256, 64), (8192, 64, 256), (8192, 52, 256)]
for lib in ('cublas', 'cublaslt'):
    torch.backends.cuda.preferred_blas_library(lib)
    out = {}
    for (m, k, n) in shapes:
        a, b = torch.randn(k, m, device=d), torch.randn(k, n, device=d)
        if m == 8192:      # forward-type: (rows, in) x (in, out)
            a, b = torch.randn(m, k, device=d), torch.randn(k, n, device=d)
            out[f'{m}x{k}x{n}'] = round(t(lambda: torch.mm(a, b)), 2)
        else:
            out[f'{m}x{k}x{n}(At)'] = round(t(lambda: torch.mm(a.t(), b)), 2)
    print(lib, json.dumps(out))

from pedestrians_video_2_carla_amd import ops
out = {}
for (m, k, n) in [(256, 8192, 64), (256, 8192, 52), (52, 8192, 64)]:
    a, b = torch.randn(k, m, device=d), torch.randn(k, n, device=d)
    c = torch.empty(m, n, device=d); cb = torch.empty(m, device=d)
    out[f'{m}x{k}x{n}'] = round(t(lambda: ops.atb(a, b, bias=True, out=c, bias_out=cb)), 2)
print('p2c_atb', json.dumps(out))
