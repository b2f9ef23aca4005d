"""Device time of the fused-MLP forward / backward at several row counts (HIP-graph of back-to-back launches)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import ops

d = torch.device('cuda:0')
dims = [52, 26, 13, 6, 39, 78, 156]
torch.manual_seed(0)
Ws = [torch.randn(o, i, device=d) * 0.1 for i, o in zip(dims[:-1], dims[1:])]
bs = [torch.randn(o, device=d) * 0.1 for o in dims[1:]]
for w in Ws + bs:
    w.requires_grad_(True)


def timeit(fn, reps=20, rounds=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(rounds):
            g.replay()
        e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * rounds)


for N in [int(a) for a in sys.argv[1:]] or [16, 256, 4096, 16384, 131072]:
    x = torch.randn(N, dims[0], device=d)
    gy = torch.randn(N, dims[-1], device=d)
    y = ops.fused_mlp(x, Ws, bs)
    sinks = [torch.zeros_like(t) for pair in zip(Ws, bs) for t in pair]

    def fwd():
        with torch.no_grad():
            ops.fused_mlp(x, Ws, bs)

    def fb():
        yy = ops.FusedMLPFunction.apply(x, len(Ws), sinks, None, False, None, 0, *Ws, *bs)
        yy.backward(gy)

    tf = timeit(fwd)
    tfb = timeit(fb)
    print(json.dumps(dict(N=N, fwd_us=round(tf, 2), bwd_us=round(tfb - tf, 2))))
