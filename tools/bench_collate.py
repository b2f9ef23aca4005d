"""Device time and HBM roofline of the K11 input-pipeline kernel (p2c_collate_fwd), graph-timed like bench.py's kernel groups.

    python tools/bench_collate.py [N ...]      # clips per launch, default 256 8192 65536

Algorithmic bytes per frame = every tensor at the boundary once: raw (Jd*C) + noise (Jd*2) + miss_u (Jd) + bboxes (4) floats
in; frames (Ji*Cf) + three targets (Ji*2 each) + shift (2) + scale (1) + bboxes (4) floats out. Prints one JSON line per N.
Also times the CPU oracle (the reference's per-clip chain, batched) on a bounded sample for the ratio.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import _graph_us
from pedestrians_video_2_carla_amd import ops
from pedestrians_video_2_carla_amd.data.base.skeleton import get_common_indices
from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
from pedestrians_video_2_carla_amd.data.openpose.skeleton import BODY_25_SKELETON

HBM_PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md


def case(N, T, device, seed=3):
    g = torch.Generator().manual_seed(seed)
    J = len(BODY_25_SKELETON)
    raw = torch.rand(N, T, J, 3, generator=g) * torch.tensor([600., 400., 0.9]) + torch.tensor([100., 50., 0.05])
    raw[torch.rand(N, T, J, generator=g) < 0.05] = 0.0
    boxes = torch.stack((raw[..., :2].amin(-2) - 4, raw[..., :2].amax(-2) + 4), -2)
    dst, src = get_common_indices(input_nodes=BODY_25_SKELETON, output_nodes=CARLA_SKELETON)
    kw = dict(flip_perm=BODY_25_SKELETON.get_flip_mask(), is_flipped=(torch.rand(N, generator=g) < 0.5).to(torch.uint8),
              rotation=(torch.rand(N, generator=g) * 2 - 1) * 10, bboxes=boxes,
              clip_size=torch.tensor([[1920., 1080.]]).repeat(N, 1), noise=torch.randn(N, T, J, 2, generator=g),
              miss_u=torch.rand(N, T, J, generator=g), miss_prob=[0.1] * J, transform='hips_neck_bbox',
              hips_idx=(BODY_25_SKELETON.MidHip.value,), neck_idx=(BODY_25_SKELETON.Neck.value,),
              src_idx=list(src), dst_idx=list(dst), n_input_joints=26)
    to = lambda v: v.to(device) if isinstance(v, torch.Tensor) else v
    return to(raw), {k: to(v) for k, v in kw.items()}, raw, kw


def main():
    device = torch.device('cuda:0')
    T, Jd, Ji, C, Cf = 16, 25, 26, 3, 2
    per_frame = 4 * (Jd * C + Jd * 2 + Jd + 4 + Ji * Cf + 3 * Ji * 2 + 2 + 1 + 4)
    stream = torch.cuda.Stream(device)
    for N in [int(a) for a in sys.argv[1:]] or [256, 8192, 65536]:
        raw, kw, raw_cpu, kw_cpu = case(N, T, device)
        with torch.cuda.stream(stream):
            us = _graph_us(lambda: ops.collate(raw, **kw), stream)
        alg = per_frame * N * T
        out = {'kernel': 'collate_kernel<32>', 'clips': N, 'T': T, 'us_per_launch': round(us, 2),
               'clips_per_s': round(N / us * 1e6), 'algorithmic_bytes': alg,
               'roofline': {'bound': 'hbm', 'achieved': round(alg / us / 1e3, 1), 'peak': HBM_PEAK, 'unit': 'GB/s',
                            'frac': round(alg / us / 1e3 / HBM_PEAK, 4)}}
        if N <= 8192:
            from oracle import collate as OC       # CPU baseline only (bench leg), never on the product path
            n = min(N, 2048)
            sub = {k: (v[:n] if isinstance(v, torch.Tensor) and v.shape[:1] == (N,) else v) for k, v in kw_cpu.items()}
            sub['flip_mask'], sub['hips'], sub['neck'] = sub.pop('flip_perm'), sub.pop('hips_idx'), sub.pop('neck_idx')
            sub['miss_prob'] = torch.tensor(sub['miss_prob'])
            t0 = time.perf_counter()
            OC.collate(raw_cpu[:n], **sub)
            dt = time.perf_counter() - t0
            out['cpu_port'] = {'clips_per_s': round(n / dt), 'threads': torch.get_num_threads(), 'sample': f'{n} clips'}
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
