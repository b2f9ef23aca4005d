#!/usr/bin/env python3
"""bench.py -- clips/sec of the pose_lifting train step (LinearAE, loc_2d_3d) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = on_train_batch_start + training_step (LinearAE forward, HIP pose head forward) + backward (HIP pose head
backward, LinearAE backward) + [one flat RCCL all-reduce of the gradients] + AdamW, on a synthetic CarlaRecorded-shaped
batch that is resident in HBM before the timed region. Workload at N=1: BASELINE.json's metric configuration
(B=256 clips per GPU, T=16, J=26, pose_changes output, loss loc_2d_3d); weak scaling: every rank gets its own B clips.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  step_breakdown  device time of every launch group of the step at the benchmark batch (graph-timed, HIP events)
  roofline     the launch group that takes the most time in the step: algorithmic bytes (pose head, vs 8 TB/s HBM) or
               flops (fused MLP, vs the 157.3 TFLOP/s fp32 MFMA peak) per launch / measured duration; `other` = the rest
  roofline_sweep  the pose-head kernels at B = 256, 1024, 8192, 65536 (the step at B=256 is latency-bound by construction)
  cpu_baseline the op-for-op CPU port of the reference step (oracle/reference_port.py) timed on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD (= the fp32 vector peak)
T_FRAMES, JOINTS = 16, 26
# SURVEY.md §8d algorithmic bytes per clip (T=16, J=26), pose_changes: fwd reads y6d(6)+gt2d(2)+gt3d(3) floats per
# joint-frame; bwd reads the same and writes grad_y(6); + the skeleton-type index.
BYTES_FWD = 4 * T_FRAMES * JOINTS * 11 + 4
BYTES_BWD = 4 * T_FRAMES * JOINTS * 17 + 4


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=200)
    p.add_argument('--warmup', type=int, default=20)
    p.add_argument('--batch-size', type=int, default=256, help='clips per GPU')
    p.add_argument('--no-graph', action='store_true', help='eager launches instead of HIP-graph replay')
    p.add_argument('--full-outputs', action='store_true', help='materialise the logging tensors in training_step')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-sweep', action='store_true')
    p.add_argument('--cpu-seconds', type=float, default=15.0)
    return p.parse_args()


def build_step(device, batch_size, use_graph, lean):
    from pedestrians_video_2_carla_amd.data.carla.carla_recorded_synthetic import SyntheticCarlaRecordedDataModule
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.pose_lifting import LitPoseLiftingFlow
    from pedestrians_video_2_carla_amd.modules.movements.linear_ae import LinearAE
    from pedestrians_video_2_carla_amd.trainer import Trainer, seed_everything
    seed_everything(22742)                                     # same init on every rank (+ rank-0 broadcast)
    dm = SyntheticCarlaRecordedDataModule(clip_length=T_FRAMES, batch_size=batch_size)
    model = LinearAE(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON)
    flow = LitPoseLiftingFlow(movements_model=model, loss_modes=['loc_2d_3d'], transform='hips_neck_bbox',
                              lean_train_outputs=lean)
    trainer = Trainer(device=device, use_graph=use_graph).setup(flow, dm)
    rank = dist.get_rank() if dist.is_initialized() else 0
    batch = dm.generate_batch(device, seed_offset=rank)        # seed 22742 + rank, staged on device once
    return flow, dm, trainer, batch


def kernel_times(device, B, reps=20):
    """Device time of the pose-head forward and backward kernels at batch B: a HIP graph of `reps` back-to-back
    launches on the launch stream, bracketed by HIP events (no host launch gaps inside the measurement)."""
    import ctypes
    from pedestrians_video_2_carla_amd import _lib, ops
    lib = _lib.lib()
    g = torch.Generator(device=device).manual_seed(1)
    y = torch.randn(B, T_FRAMES, JOINTS, 6, device=device, generator=g)
    y[..., 0] += 1.5
    y[..., 4] += 1.5
    st = torch.randint(0, 4, (B,), device=device, generator=g).int()
    gt2 = torch.randn(B, T_FRAMES, JOINTS, 2, device=device, generator=g)
    gt3 = torch.randn(B, T_FRAMES, JOINTS, 3, device=device, generator=g)
    spec = ops.PoseHeadSpec(kind='pose_changes_6d')
    f32 = dict(dtype=torch.float32, device=device)
    bufs = {'partials': torch.empty(lib.p2c_pose_head_workspace_floats(B), **f32), 'loss_sums': torch.empty(4, **f32),
            'losses': torch.empty(3, **f32), 'final_rel_rot': torch.empty(B, JOINTS, 3, 3, **f32)}
    desc = ops._fill_desc(spec, y, st, None, None, gt2, gt3, bufs, {})
    gl = torch.tensor([0.0, 0.0, 1.0], **f32)
    gy = torch.empty_like(y)
    out = {}
    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        s = stream.cuda_stream

        def fwd():
            _lib.check(lib.p2c_pose_head_fwd(ctypes.byref(desc), s), 'fwd')

        def bwd():
            _lib.check(lib.p2c_pose_head_bwd(ctypes.byref(desc), _lib.grad_loss_pointers(vector=gl.data_ptr()), None, None, None, gy.data_ptr(), s), 'bwd')

        def train():                # what the trainer's step runs (ops.deferred_loss_finalize mode 2): the forward call only
            desc.defer_loss_finalize = 2      # counts target pairs, the backward kernel produces losses + gradients, + finalize
            fwd()
            bwd()
            desc.defer_loss_finalize = 0

        fwd()
        bwd()
        stream.synchronize()
        for name, fn in (('fwd', fwd), ('bwd', bwd), ('train', train)):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for _ in range(reps):
                    fn()
            graph.replay()
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            rounds = 5
            e0.record(stream)
            for _ in range(rounds):
                graph.replay()
            e1.record(stream)
            e1.synchronize()
            out[name] = e0.elapsed_time(e1) * 1e3 / (reps * rounds)      # us per launch (fwd: head + 1-block reduce)
    return out


def _graph_us(fn, stream, reps=20, rounds=5):
    fn()
    stream.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        for _ in range(reps):
            fn()
    graph.replay()
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(rounds):
        graph.replay()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * rounds)


def mlp_times(device, model, B, fused_update):
    """Device time of the fused LinearAE launches at N = B * T frames, through the C ABI, graph-timed like
    kernel_times(): forward (the optimizer keeps the weight image current: no pack launch), backward + partial reduction
    (+ AdamW when the single-GPU trainer fuses the optimizer step into the reduction), stand-alone AdamW otherwise.
    The weights / gradients / moments are views of flat buffers, as in the trainer."""
    import ctypes
    from pedestrians_video_2_carla_amd import _lib, ops
    from pedestrians_video_2_carla_amd.parallel.optim import FlatAdamW
    lib = _lib.lib()
    shapes = [(l.weight.shape[0], l.weight.shape[1]) for l in model._linears()]
    dims = [shapes[0][1]] + [o for o, _ in shapes]
    n = sum(o * (i + 1) for o, i in shapes)
    flat = torch.nn.Parameter(torch.randn(n, device=device) * 0.1)
    flat.grad = torch.zeros_like(flat)
    Ws, bs, gW, gb, off = [], [], [], [], 0
    for o, i in shapes:
        Ws.append(flat.data[off:off + o * i].view(o, i)), gW.append(flat.grad[off:off + o * i].view(o, i))
        off += o * i
        bs.append(flat.data[off:off + o]), gb.append(flat.grad[off:off + o])
        off += o
    N = B * T_FRAMES
    x = torch.randn(N, dims[0], device=device)
    gy = torch.randn(N, dims[-1], device=device)
    desc = ops._mlp_desc(x, Ws, bs)
    f32 = dict(dtype=torch.float32, device=device)
    y = torch.empty(N, dims[-1], **f32)
    n_image, index = ops.mlp_image_layout(dims)
    image = torch.empty(n_image, **f32)
    part = torch.empty(lib.p2c_mlp_workspace_floats(ctypes.byref(desc)), **f32)
    desc.y, desc.w_image, desc.gy, desc.partials = y.data_ptr(), image.data_ptr(), gy.data_ptr(), part.data_ptr()
    n_saved = lib.p2c_mlp_saved_floats(ctypes.byref(desc))        # many tiles per workgroup: activations saved, as in training
    saved = torch.empty(max(n_saved, 1), **f32)
    if n_saved > 0:
        desc.saved = saved.data_ptr()
    for i in range(len(Ws)):
        desc.gW[i], desc.gb[i] = gW[i].data_ptr(), gb[i].data_ptr()
    opt = FlatAdamW([flat], lr=1e-4, zero_grad_in_step=False)
    opt.set_scatter(index.to(device), image)
    opt.sync_hyper()
    out = {}
    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        s = stream.cuda_stream
        _lib.check(lib.p2c_mlp_pack(ctypes.byref(desc), s), 'mlp pack')     # in the step the optimizer keeps the image current
        desc.skip_pack = 1
        out['mlp_fwd'] = _graph_us(lambda: _lib.check(lib.p2c_mlp_fwd(ctypes.byref(desc), s), 'mlp fwd'), stream)
        bwd_name = 'mlp_bwd(+reduce)'
        if fused_update:
            od = opt.descriptor_for_fusion()
            desc.fused_adamw = ctypes.addressof(od)
            bwd_name = 'mlp_bwd(+reduce+adamw)'
        out[bwd_name] = _graph_us(lambda: _lib.check(lib.p2c_mlp_bwd(ctypes.byref(desc), s), 'mlp bwd'), stream)
        if not fused_update:
            out['adamw'] = _graph_us(opt.step, stream)
    macs = sum(o * i for o, i in shapes)
    macs_bwd = macs + sum(o * i for o, i in shapes[1:])        # wgrad of every layer + dgrad of layers 1..L-1
    return out, {'mlp_fwd': 2 * macs * N, bwd_name: 2 * macs_bwd * N}


def mfma_entry(name, B, us, flops):
    achieved = flops / (us * 1e-6) / 1e12
    return {'kernel': name, 'B': B, 'us_per_launch': round(us, 2), 'bound': 'mfma', 'achieved': round(achieved, 3),
            'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
            'algorithmic_flops_per_launch': flops, 'traffic': None}


def roofline_entry(name, B, us, nbytes_per_clip, traffic=None):
    achieved = nbytes_per_clip * B / (us * 1e-6) / 1e9
    return {'kernel': name, 'B': B, 'us_per_launch': round(us, 2), 'bound': 'hbm', 'achieved': round(achieved, 1),
            'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBPS, 4),
            'algorithmic_bytes_per_launch': nbytes_per_clip * B, 'traffic': traffic}


def load_traffic():
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return {}


def cpu_baseline(batch_size, seconds):
    """Op-for-op CPU port of the reference step on this host's cores (bounded sample of the same workload)."""
    import copy
    from oracle import pose_head as O
    from oracle import reference_port as P
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.movements.linear_ae import LinearAE
    # the GPU box gives a 1-GPU job a share of 16 host cores; hundreds of intra-op threads on ~10 KB tensors only
    # add synchronisation cost (256 threads: 131 s/step measured, vs 0.45 s/step with 8)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    torch.manual_seed(22742)
    model = LinearAE(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON)
    model.rotation_output_format = 'rotation_6d'
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-8)
    b = O.synthetic_batch(batch_size, T_FRAMES, seed=22742)
    targets = {'projection_2d_transformed': b['projection_2d_transformed'], 'absolute_pose_loc': b['absolute_pose_loc']}
    meta = {'age': b['age'], 'gender': b['gender']}
    t0 = time.perf_counter()
    P.port_train_step(model, opt, b['frames'], targets, meta)          # warm-up
    if time.perf_counter() - t0 > seconds:                              # pathological host: report the single step
        dt = time.perf_counter() - t0
        return {'value': round(batch_size / dt, 1), 'unit': 'clips/s', 'cores': cores, 'kind': 'port',
                'ms_per_step': round(dt * 1e3, 1), 'sample': f'1 (cold) step of B={batch_size},T={T_FRAMES}'}
    n, t0 = 0, time.perf_counter()
    while True:
        P.port_train_step(model, opt, b['frames'], targets, meta)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 200:
            break
    return {'value': round(batch_size * n / dt, 1), 'unit': 'clips/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'ms_per_step': round(dt / n * 1e3, 1),
            'sample': f'{n} steps of B={batch_size},T={T_FRAMES} (op-for-op port of the reference step, eager PyTorch '
                      f'CPU fp32, {torch.get_num_threads()} threads)'}


def main():
    args = parse()
    from pedestrians_video_2_carla_amd.trainer import init_distributed
    info = init_distributed()
    world, rank, local_rank = info['world_size'], info['rank'], info['local_rank']
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP hot path has no CPU fallback')
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    flow, dm, trainer, batch = build_step(device, args.batch_size, not args.no_graph, not args.full_outputs)
    for i in range(max(args.warmup, 1)):
        trainer.train_step(flow, batch, i)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = trainer.train_step(flow, batch, i)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    global_batch = args.batch_size * world
    value = global_batch * args.steps / elapsed
    result = {
        'metric': 'clips/sec (B=256,T=16,J=26) pose_lifting train step',
        'value': round(value, 1), 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'flow=pose_lifting movements_model=LinearAE(pose_changes) loss=loc_2d_3d '
                               f'transform=hips_neck_bbox clip_length={T_FRAMES} J={JOINTS} batch_size='
                               f'{args.batch_size}/GPU (BASELINE.json metric config; configs[1]/[3] = --batch-size 1024)',
                   'global_batch': global_batch, 'per_gpu_batch': args.batch_size, 'parallelism': f'dp{world}',
                   'hip_graph': not args.no_graph, 'lean_train_outputs': not args.full_outputs,
                   'deferred_loss_finalize': os.environ.get('P2C_DEFER_FINALIZE', '1') != '0',
                   'grad_allreduce_bytes': trainer.flat.nbytes(), 'final_loss': final_loss},
    }
    traffic = load_traffic()
    kt = kernel_times(device, args.batch_size)
    names = {'fwd': 'pose_head_rot_fwd<6D>(+loss_finalize)', 'bwd': 'pose_head_rot_bwd<6D>'}
    per_clip = {'fwd': BYTES_FWD, 'bwd': BYTES_BWD}
    entries = {names[w]: roofline_entry(names[w], args.batch_size, kt[w], per_clip[w],
                                        (traffic.get(f'{names[w]}@B{args.batch_size}') or {}).get('bytes')) for w in ('fwd', 'bwd')}
    if int(os.environ.get('P2C_DEFER_FINALIZE', '2')) == 2 and args.batch_size <= 2048:
        breakdown = {'pose_head_train(count + fwd/bwd in one kernel + finalize)': round(kt['train'], 2)}
    else:
        breakdown = {names[w]: round(kt[w], 2) for w in ('fwd', 'bwd')}
    if getattr(flow.movements_model, 'fused_mlp', False):
        mt, flops = mlp_times(device, flow.movements_model, args.batch_size, getattr(trainer, '_opt_in_backward', False))
        breakdown.update({k: round(v, 2) for k, v in mt.items()})
        for k, fl in flops.items():
            entries[k] = mfma_entry(k, args.batch_size, mt[k], fl)
            entries[k]['traffic'] = (traffic.get(f'{k}@B{args.batch_size}') or {}).get('bytes')
    result['step_breakdown_us'] = breakdown
    dominant = max(entries, key=lambda k: entries[k]['us_per_launch'])
    result['roofline'] = entries.pop(dominant)
    result['roofline']['other'] = list(entries.values())
    if not args.no_sweep and world == 1:
        sweep = []
        for B in (256, 1024, 8192, 65536):
            k = kernel_times(device, B, reps=10 if B > 8192 else 20)
            for which in ('fwd', 'bwd'):
                sweep.append(roofline_entry(names[which], B, k[which], per_clip[which],
                                            (traffic.get(f'{names[which]}@B{B}') or {}).get('bytes')))
        result['roofline_sweep'] = sweep
    if not args.no_cpu_baseline and world == 1:
        result['cpu_baseline'] = cpu_baseline(args.batch_size, args.cpu_seconds)
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
