"""Kernel micro-benchmark: pose head fwd / bwd device time per launch (HIP graph of 20 launches, events on the launch
stream) for both kernel variants at several batch sizes.   python tools/kbench.py [B ...]"""
import sys, os, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import ops, _lib

FWD_B, BWD_B = 18308, 28292   # algorithmic bytes per clip at T=16
T, J = 16, 26


def graph_time(fn, stream, reps=20, rounds=5):
    fn(); stream.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for _ in range(reps):
            fn()
    g.replay(); stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(rounds):
        g.replay()
    e1.record(stream); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * rounds)


def main():
    d = torch.device('cuda:0')
    lib = _lib.lib()
    Bs = [int(a) for a in sys.argv[1:]] or [256, 1024, 2048, 4096, 8192]
    stream = torch.cuda.Stream()
    f32 = dict(dtype=torch.float32, device=d)
    for B in Bs:
        g = torch.Generator(device=d).manual_seed(1)
        y = torch.randn(B, T, J, 6, device=d, generator=g)
        y[..., 0] += 1.5; y[..., 4] += 1.5
        st = torch.randint(0, 4, (B,), device=d, generator=g).int()
        gt2 = torch.randn(B, T, J, 2, device=d, generator=g)
        gt3 = torch.randn(B, T, J, 3, device=d, generator=g)
        spec = ops.PoseHeadSpec(kind='pose_changes_6d')
        bufs = {'partials': torch.empty(lib.p2c_pose_head_workspace_floats(B), **f32), 'loss_sums': torch.empty(4, **f32),
                'losses': torch.empty(3, **f32), 'final_rel_rot': torch.empty(B, J, 3, 3, **f32)}
        desc = ops._fill_desc(spec, y, st, None, None, gt2, gt3, bufs, {})
        gl = torch.tensor([0.0, 0.0, 1.0], **f32)
        gy = torch.empty_like(y)
        variants = (('time_parallel', 1 << 30, 1 << 30, 1 << 30), ('joint_lane', 0, 1 << 30, 1 << 30), ('packed', 0, 0, 1 << 30),
                    ('chain_lane', 0, 1 << 30, 0))
        only = os.environ.get('KBENCH_VARIANTS')
        for variant, max_b, min_pk, min_chain in variants:
            if only and variant not in only.split(','):
                continue
            if variant == 'time_parallel' and B > 8192:
                continue
            prev = lib.p2c_pose_head_set_time_parallel_max_batch(max_b)
            prev_pk = lib.p2c_pose_head_set_packed_min_batch(min_pk)
            prev_ch = lib.p2c_pose_head_set_chain_min_batch(min_chain)
            with torch.cuda.stream(stream):
                s = stream.cuda_stream
                tf = graph_time(lambda: _lib.check(lib.p2c_pose_head_fwd(ctypes.byref(desc), s), 'fwd'), stream)
                tb = graph_time(lambda: _lib.check(lib.p2c_pose_head_bwd(
                    ctypes.byref(desc), _lib.grad_loss_pointers(vector=gl.data_ptr()), None, None, None, gy.data_ptr(), s), 'bwd'),
                    stream)
            lib.p2c_pose_head_set_time_parallel_max_batch(prev)
            lib.p2c_pose_head_set_packed_min_batch(prev_pk)
            lib.p2c_pose_head_set_chain_min_batch(prev_ch)
            print(json.dumps(dict(B=B, variant=variant, fwd_us=round(tf, 2), bwd_us=round(tb, 2),
                                  fwd_GBps=round(FWD_B * B / tf / 1e3, 1), bwd_GBps=round(BWD_B * B / tb / 1e3, 1),
                                  fwd_frac=round(FWD_B * B / tf / 1e3 / 8000, 3), bwd_frac=round(BWD_B * B / tb / 1e3 / 8000, 3))), flush=True)


if __name__ == '__main__':
    main()
