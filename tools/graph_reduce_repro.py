"""Pure-torch check: is the gradient of a broadcast parameter (a reduction over many rows inside backward) still right when a
captured step is replayed after the allocator has handed out and taken back other memory?"""
import torch
d = torch.device('cuda:0')
torch.manual_seed(0)
R, N = 546624 // 4, 832
x = torch.randn(R, N, device=d)
p = torch.zeros(1, N, device=d, requires_grad=True)
w = torch.ones(9, device=d, requires_grad=True)
b = torch.zeros(1, device=d, requires_grad=True)
z = torch.randn(2336, 9, 832, device=d)
p.grad, w.grad, b.grad = torch.zeros_like(p), torch.zeros_like(w), torch.zeros_like(b)


def step():
    for t in (p, w, b):
        t.grad.zero_()
    y = ((x + p) ** 2).mean() + ((z * w.view(1, -1, 1)).sum(1) + b).pow(2).mean()
    y.backward()
    return y.detach()


ref = step()
want = [t.grad.clone() for t in (p, w, b)]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    out = step()
for i in range(4):
    g.replay()
    torch.cuda.synchronize()
    errs = [float((t.grad - v).abs().max() / v.abs().max()) for t, v in zip((p, w, b), want)]
    print('replay', i, 'relative errors of the three gradients', ['%.2e' % e for e in errs], flush=True)
    junk = [torch.empty(1 << 20, device=d).normal_() for _ in range(64)]
    del junk
