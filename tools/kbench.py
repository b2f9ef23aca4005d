"""Kernel micro-benchmark: pose head fwd / bwd device time at several batch sizes (HIP events on the launch stream)."""
import sys, os, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import ops, _lib

FWD_B, BWD_B = 18308, 28292   # algorithmic bytes per clip at T=16


def time_it(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    d = torch.device('cuda:0')
    kind = sys.argv[1] if len(sys.argv) > 1 else 'pose_changes_6d'
    spec = ops.PoseHeadSpec(kind=kind)
    for B in (256, 1024, 8192, 65536):
        T = 16
        g = torch.Generator(device=d).manual_seed(1)
        ny = {'pose_changes_6d': (6,), 'pose_changes': (3, 3), 'absolute_loc': (3,)}[kind]
        y = torch.randn((B, T, 26) + ny, device=d, generator=g)
        st = torch.randint(0, 4, (B,), device=d, generator=g).int()
        gt2 = torch.randn(B, T, 26, 2, device=d, generator=g)
        gt3 = torch.randn(B, T, 26, 3, device=d, generator=g)
        yr = y.clone().requires_grad_(True)
        losses, _ = ops.pose_head(yr, spec, st, gt2d=gt2, gt3d=gt3)
        gl = torch.tensor([0., 0., 1.], device=d)

        def fwd():
            ops.pose_head(y, spec, st, gt2d=gt2, gt3d=gt3)

        def fb():
            l, _ = ops.pose_head(yr, spec, st, gt2d=gt2, gt3d=gt3)
            torch.autograd.backward(l, gl)

        tf = time_it(fwd)
        tfb = time_it(fb)
        tb = tfb - tf
        print(json.dumps(dict(kind=kind, B=B, fwd_us=round(tf, 1), fwd_bwd_us=round(tfb, 1),
                              fwd_GBps=round(FWD_B * B / tf / 1e3, 1), bwd_GBps=round(BWD_B * B / max(tb, 1e-3) / 1e3, 1),
                              clips_per_s=round(B / tfb * 1e6))))


if __name__ == '__main__':
    main()
