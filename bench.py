#!/usr/bin/env python3
"""bench.py -- clips/sec of the pose_lifting train step (LinearAE, loc_2d_3d) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = training_step (LinearAE forward, HIP pose head forward) + backward (HIP pose head backward, LinearAE backward)
+ [one flat RCCL all-reduce of the gradients] + AdamW, on a synthetic CarlaRecorded-shaped batch that is resident in HBM
before the timed region (its per-batch constants -- skeleton-type index, target-pair counts: on_train_batch_start -- are
derived when the batch is staged). Workload at N=1: BASELINE.json's metric configuration (B=256 clips per GPU, T=16, J=26,
pose_changes output, loss loc_2d_3d); weak scaling: every rank gets its own B clips. On one GPU at this batch the step is
two launches (csrc/p2c_train.hip); the trainer captures it, verifies that the captured graph holds exactly the two kernels
of its one recorded C-ABI call, and then replays the step by making that call (config.direct_replay: a graph launch costs
~5 us of start-up per replay).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  repeat_ms_per_step  p50 / min / max of five more timed blocks of K steps (one hiccup cannot move the headline unseen)
  fresh_batch_ms_per_step  the same step fed a NEW batch object every iteration: staging copy + batch-start hook + replay
  step_breakdown_us  device time of every launch of the step at the benchmark batch (graph-timed, HIP events)
  roofline     the launch that takes the most time in the step: algorithmic flops (vs the 157.3 TFLOP/s fp32 MFMA peak) or
               bytes (vs 8 TB/s HBM) per launch / measured duration, PMC traffic; `other` = the rest
  roofline_sweep  the pose-head kernels p2c_pose_head_fwd / _bwd dispatch to at B = 256, 1024, 8192, 16384, 65536 (time-parallel,
                  joint-lane, chain-lane by batch; each kernel timed alone, the forward op with its loss reduction beside it)
  step_sweep   the FULL train step at B = 256, 1024, 8192, 65536: ms per step, clips/s, the fraction of the fp32-MFMA peak (1.61
               MFLOP per clip) and of the HBM peak (11 652 B per clip) that rate is, the step's launches one by one, and which
               form of the fused step ran (a workgroup per clip / a pair of wavefronts per clip; weight gradient per tile / streamed)
  extra_configs  BASELINE.json configs[1] (B = 1024), configs[2] (autoencoder, Seq2SeqEmbeddings, B = 512) and one GPU's share
                 of configs[4] (PoseFormer, clip_length 81, B = 32) on this GPU
  cpu_baseline the op-for-op CPU port of the reference step (oracle/reference_port.py) timed on this host's cores:
               all cores, one thread, and under torch DDP / gloo with world_size 1, 2, 4, 8 (oracle/ddp_baseline.py).
"""
import argparse
import json
import os
import statistics
import subprocess
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
    p.add_argument('--cpu-seconds', type=float, default=12.0)
    p.add_argument('--no-extra-configs', action='store_true')
    p.add_argument('--repeats', type=int, default=5)
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

        def fwd():                  # the forward op: the pose-head kernel + the one-workgroup loss reduction behind it
            _lib.check(lib.p2c_pose_head_fwd(ctypes.byref(desc), s), 'fwd')

        def fwd_kernel():           # the pose-head kernel alone (what rocprofv3's per-kernel average is compared with)
            _lib.check(lib.p2c_pose_head_fwd_launch(ctypes.byref(desc), 1, s), 'fwd kernel')

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
        for name, fn in (('fwd_op', fwd), ('fwd', fwd_kernel), ('bwd', bwd), ('train', train)):
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
            out[name] = e0.elapsed_time(e1) * 1e3 / (reps * rounds)      # us per launch (fwd_op: head + 1-block reduce)
    return out


def head_kernel_names(B):
    """The pose-head kernels p2c_pose_head_fwd / _bwd dispatch to at batch B (6-D kind, lean outputs, T = 16): time-parallel up
    to P2C_TP_MAX_B (2048), chain-lane from P2C_CHAIN_MIN_B (8192), joint-lane clip-sequential between."""
    tp, ch = int(os.environ.get('P2C_TP_MAX_B', '2048')), int(os.environ.get('P2C_CHAIN_MIN_B', '8192'))
    if B <= tp:
        return {'fwd': 'pose_head_rot_fwd_tp<6D>', 'bwd': 'pose_head_rot_bwd_tangent_tp<6D>'}
    if B >= ch:
        return {'fwd': 'pose_head_chain_fwd<6D>', 'bwd': 'pose_head_chain_bwd<6D>'}
    return {'fwd': 'pose_head_rot_fwd<6D>', 'bwd': 'pose_head_rot_bwd_tangent<6D>'}


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


def fused_step_times(device, flow, trainer, batch):
    """Device time of each launch of the two-launch train step (csrc/p2c_train.hip) on the bench's own flow, weights and
    batch, through the C ABI's measurement hook p2c_train_step_launch: HIP graphs of 20 back-to-back launches on the launch
    stream, bracketed by HIP events. Returns ({name: us}, {name: algorithmic flops per launch}) or None when the flow does
    not take the fused step at this batch. (The optimizer really steps: call after the loss has been read.)"""
    import ctypes
    from pedestrians_video_2_carla_amd import _lib, ops
    lib = _lib.lib()
    frames, targets, _meta = trainer._static_batch if trainer._static_batch is not None else batch
    flow.train()
    plan = flow._fused_train_plan(frames, targets) if hasattr(flow, '_fused_train_plan') else None
    counts = getattr(flow, '_pair_counts', None)
    if plan is None or counts is None:
        return None
    spec, gt2d, gt3d = plan
    fa = flow.movements_model.fused_args(device)
    f32 = dict(dtype=torch.float32, device=device)
    B, T = frames.shape[:2]
    bufs = {'partials': torch.empty(B * 4, **f32), 'loss_sums': torch.empty(4, **f32), 'losses': torch.empty(3, **f32)}
    sinks = fa['sinks']
    if sinks is None:
        return None
    desc, keep = ops.train_step_desc(frames, spec, flow.projection._skel_type, None, None, gt2d, gt3d, counts, fa['weights'],
                                     fa['biases'], sinks[0::2], sinks[1::2], bufs, fa['image'], fa['image_is_current'],
                                     fa['fused_optimizer'])
    gl = torch.tensor([0.0, 0.0, 1.0], **f32)
    out = {}
    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        s = stream.cuda_stream
        glp = _lib.grad_loss_pointers(vector=gl.data_ptr())
        for name, which in (('train_clip_kernel', 1), ('train_wgrad_kernel(+adamw+loss)', 2), ('train_step(2 launches)', 3)):
            _lib.check(lib.p2c_train_step_launch(ctypes.byref(desc), glp, 3, s), 'train step')      # valid factors / counters
            out[name] = _graph_us(lambda: _lib.check(lib.p2c_train_step_launch(ctypes.byref(desc), glp, which, s), name), stream)
    shapes = [(l.weight.shape[0], l.weight.shape[1]) for l in flow.movements_model._linears()]
    macs = sum(o * i for o, i in shapes)
    macs_dgrad = sum(o * i for o, i in shapes[1:])
    N = B * T
    flops = {'train_clip_kernel': 2 * (macs + macs_dgrad) * N, 'train_wgrad_kernel(+adamw+loss)': 2 * macs * N}
    return out, flops


def recurrence_times(device, B, T=T_FRAMES, H=64, O=52):
    """Device time of the Seq2SeqEmbeddings recurrences at cfg3's shapes through the C ABI (graph-timed): one LSTM layer's time
    loop forward / backward (K7b) and the T-step decoder loop forward / backward (K7c), each with its algorithmic flops
    (the h W_hh^T products of the cells; K7c adds its input projections and fc_out)."""
    import ctypes
    from pedestrians_video_2_carla_amd import _lib
    lib = _lib.lib()
    f32 = dict(dtype=torch.float32, device=device)
    g = torch.Generator(device=device).manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g, **f32) * 0.2       # noqa: E731
    keep = []
    d = _lib.LstmDesc()
    d.T, d.B, d.H = T, B, H
    ten = {k: rnd(*s) for k, s in dict(gx=(T, B, 4 * H), w_hh=(4 * H, H), out=(T, B, H), hT=(B, H), cT=(B, H), acts=(T, B, 4 * H),
                                       cs=(T, B, H), g_out=(T, B, H), g_gx=(T, B, 4 * H)).items()}
    for k, v in ten.items():
        setattr(d, k, v.data_ptr())
    keep.append(ten)
    dd = _lib.DecoderDesc()
    dd.T, dd.B, dd.H, dd.O = T, B, H, O
    # the launches the step itself issues (ops.DecoderStackFunction): the frame-invariant terms k_l = b + hid_l W_hh_l^T and their
    # gradients formed inside, the output written batch-first as well, the loss gradient read batch-first, dropout mask on
    ten2 = {k: rnd(*s) for k, s in dict(c0=(B, H), c1=(B, H), w_ih0=(4 * H, O), w_ih1=(4 * H, H),
                                        w_fc=(O, H), b_fc=(O,), out=(T, B, O), acts0=(T, B, 4 * H), acts1=(T, B, 4 * H),
                                        h0d=(T, B, H), h1=(T, B, H), g_out=(B, T, O), g_gates0=(T, B, 4 * H),
                                        g_gates1=(T, B, 4 * H), g_outtot=(T, B, O), g_c0=(B, H), g_c1=(B, H),
                                        hid0=(B, H), hid1=(B, H), w_hh0=(4 * H, H), w_hh1=(4 * H, H), b0a=(4 * H,), b0b=(4 * H,),
                                        b1a=(4 * H,), b1b=(4 * H,), kw0=(B, 4 * H), kw1=(B, 4 * H), out_bt=(B, T, O),
                                        g_k0=(B, 4 * H), g_k1=(B, 4 * H), g_hid0=(B, H), g_hid1=(B, H)).items()}
    ten2['drop'] = (torch.rand(T, B, H, generator=g, **f32) > 0.2).float() / 0.8
    for k, v in ten2.items():
        setattr(dd, k, v.data_ptr())
    dd.g_out_bt = 1
    keep.append(ten2)
    out = {}
    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        s = stream.cuda_stream
        out['lstm_rec_fwd (K7b, one layer)'] = _graph_us(lambda: _lib.check(lib.p2c_lstm_rec_fwd(ctypes.byref(d), s), 'rec fwd'), stream)
        out['lstm_rec_bwd (K7b, one layer)'] = _graph_us(lambda: _lib.check(lib.p2c_lstm_rec_bwd(ctypes.byref(d), s), 'rec bwd'), stream)
        out['decoder_loop_fwd (K7c)'] = _graph_us(lambda: _lib.check(lib.p2c_decoder_fwd(ctypes.byref(dd), s), 'dec fwd'), stream)
        out['decoder_loop_bwd (K7c)'] = _graph_us(lambda: _lib.check(lib.p2c_decoder_bwd(ctypes.byref(dd), s), 'dec bwd'), stream)
    rec = 2 * T * B * H * 4 * H                                        # h_{t-1} W_hh^T for all t
    dec = 2 * T * B * (O * 4 * H + H * 4 * H + H * O) + 2 * 2 * B * H * 4 * H     # W_ih0 x_t, W_ih1 h0_t, fc_out h1_t; the two k_l
    flops = {'lstm_rec_fwd (K7b, one layer)': rec, 'lstm_rec_bwd (K7b, one layer)': rec,
             'decoder_loop_fwd (K7c)': dec, 'decoder_loop_bwd (K7c)': dec}
    return out, flops


def gemm_times(device, rows=2336 * 9, width=832):
    """Device time of K16 at the shapes of one temporal PoseTransformer block (cfg5: 2 336 windows x 9 frame tokens, 832
    features), graph-timed through ops.gemm / ops.gemm_tn: the qkv layer forward (NT), its input gradient (NN) and its weight
    gradient (TN, both launches), each with its 2 M N K flop."""
    from pedestrians_video_2_carla_amd import ops
    g = torch.Generator(device='cpu').manual_seed(5)
    x = torch.randn(rows, width, generator=g).to(device)
    w = torch.randn(3 * width, width, generator=g).to(device)
    gy = torch.randn(rows, 3 * width, generator=g).to(device)
    y, gx, gw = torch.empty(rows, 3 * width, device=device), torch.empty(rows, width, device=device), torch.empty(3 * width, width, device=device)
    out = {}
    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        out['K16 NT qkv forward (21024 x 2496 x 832)'] = _graph_us(lambda: ops.gemm(x, w, True, out=y), stream, reps=5, rounds=3)
        out['K16 NN qkv input gradient (21024 x 832 x 2496)'] = _graph_us(lambda: ops.gemm(gy, w, False, out=gx), stream, reps=5, rounds=3)
        out['K16 TN qkv weight gradient (2496 x 832 from 21024 rows; slabs + sum)'] = _graph_us(lambda: ops.gemm_tn(gy, x, out=gw), stream, reps=5, rounds=3)
    flops = {k: 2.0 * rows * width * 3 * width for k in out}
    return out, flops


STEP_FLOPS_PER_CLIP = 2.0 * (17212 + 33072) * T_FRAMES      # LinearAE forward + dgrad + wgrad of one clip
STEP_BYTES_PER_CLIP = 4 * T_FRAMES * JOINTS * (2 + 2 + 3) + 4   # frames + both targets in, nothing out but three scalars


def step_sweep(device, sizes=(256, 512, 1024, 8192, 65536)):
    """The whole train step (trainer.train_step on a resident batch: LinearAE pose_changes, loc_2d_3d, fp32, AdamW in the step) at
    several batch sizes: ms per step from three blocks of wall-clocked steps (min), clips/s, the fraction of the fp32-MFMA peak
    (1.61 MFLOP per clip) and of the HBM peak (11 652 B per clip) that rate corresponds to, and the step's launches, each timed
    through p2c_train_step_launch. `forms` names which form of the two launches ran (latency / throughput)."""
    from pedestrians_video_2_carla_amd import _lib
    lib = _lib.lib()
    stream_min, wgrad_min = lib.p2c_train_step_set_stream_min_batch(-1), lib.p2c_train_step_set_wgrad_stream_min_batch(-1)
    out = []
    for Bs in sizes:
        flow, dm, trainer, batch = build_step(device, Bs, True, True)
        steps = max(10, min(200, (1 << 21) // Bs))
        for i in range(10):
            trainer.train_step(flow, batch, i)
        torch.cuda.synchronize(device)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(steps):
                trainer.train_step(flow, batch, i)
            torch.cuda.synchronize(device)
            ts.append((time.perf_counter() - t0) / steps)
        sec = min(ts)
        rate = Bs / sec
        entry = {'B': Bs, 'ms_per_step': round(sec * 1e3, 4), 'clips_per_s': round(rate, 1),
                 'frac_mfma_f32': round(rate * STEP_FLOPS_PER_CLIP / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                 'frac_hbm': round(rate * STEP_BYTES_PER_CLIP / 1e9 / HBM_PEAK_GBPS, 5),
                 'two_launch_step': getattr(flow, '_pair_counts', None) is not None}
        fused = fused_step_times(device, flow, trainer, batch)
        if fused is not None:
            entry['launch_us'] = {k: round(v, 2) for k, v in fused[0].items()}
            traffic = load_traffic()
            entry['traffic'] = {k: (traffic.get(f'{k}@B{Bs}') or {}).get('bytes')      # HBM bytes per launch (PMC passes, profiles/traffic.json)
                                for k in ('train_clip_kernel', 'train_wgrad_kernel(+adamw+loss)')}
            entry['algorithmic_bytes'] = STEP_BYTES_PER_CLIP * Bs
            entry['forms'] = {'first_launch': 'train_stream_kernel (a pair of wavefronts per clip)' if Bs >= stream_min
                              else 'train_clip_kernel (a workgroup per clip)',
                              'second_launch': 'wgrad_stream_kernel + wgrad_reduce_kernel' if Bs >= wgrad_min
                              else 'train_wgrad_kernel (combine, AdamW and losses in the launch)'}
        out.append(entry)
        del flow, dm, trainer, batch
        torch.cuda.empty_cache()
    return out


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
    """profiles/traffic.json (PMC passes, tools/make_traffic.py), minus the entries measured on kernel sources that have changed
    since (their stamp, tools/traffic_stamp.py, no longer matches the tree): a stale figure is dropped, not printed."""
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        stored = json.load(f)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        from traffic_stamp import src_sha16
    except Exception:                                                   # noqa: BLE001 -- no stamps: nothing can be trusted
        return {}
    finally:
        sys.path.pop(0)
    return {k: v for k, v in stored.items() if v.get('src_sha16') is not None and v.get('src_sha16') == src_sha16(k)}


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_ddp_baseline(cores, seconds):
    """The CPU port under torch DDP / gloo, world_size 1, 2, 4, 8, threads = cores / world_size."""
    out = {}
    script = os.path.join(ROOT, 'oracle', 'ddp_baseline.py')
    # torch's autograd engine opens the GPU in every process that runs a backward (its device-thread set-up asks the HIP
    # runtime for a device count), and a GPU box admits at most 6 processes on its card: with this process holding it too,
    # world_size 8 cannot run there. P2C_CPU_DDP_MAX_WORLD=8 on a host without that guard.
    max_world = int(os.environ.get('P2C_CPU_DDP_MAX_WORLD', '4'))
    for world in (1, 2, 4, 8):
        if world > cores:
            break
        if world > max_world:
            out[str(world)] = {'skipped': f'at most {max_world} ranks beside the benchmark process on this box (process guard: 6 '
                                          f'processes per GPU; every rank\'s autograd engine opens the device)'}
            continue
        threads = max(1, cores // world)
        port = 29650 + world
        procs = []
        try:
            for r in range(world):
                env = dict(os.environ, OMP_NUM_THREADS=str(threads), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                           RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), HIP_VISIBLE_DEVICES='',
                           CUDA_VISIBLE_DEVICES='')
                procs.append(subprocess.Popen([sys.executable, script, '--seconds', str(seconds), '--threads', str(threads)],
                                              env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                              stderr=subprocess.DEVNULL, text=True))
            stdout, _ = procs[0].communicate(timeout=seconds * 6 + 120)
            for pr in procs[1:]:
                pr.wait(timeout=60)
            line = [l for l in stdout.splitlines() if l.startswith('{')]
            out[str(world)] = json.loads(line[-1]) if line else {'error': 'no result line'}
        except Exception as e:                                      # noqa: BLE001 -- a baseline leg must not sink the bench
            out[str(world)] = {'error': repr(e)[:200]}
            for pr in procs:                                        # exactly the processes started here
                if pr.poll() is None:
                    pr.kill()
    return out


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
    result = {'value': round(batch_size * n / dt, 1), 'unit': 'clips/s', 'cores': torch.get_num_threads(), 'kind': 'port',
              'ms_per_step': round(dt / n * 1e3, 1),
              'sample': f'{n} steps of B={batch_size},T={T_FRAMES} (op-for-op port of the reference step, eager PyTorch '
                        f'CPU fp32, {torch.get_num_threads()} threads)',
              'cpu_model': _cpu_model(), 'host_cores': os.cpu_count(),
              'note': 'per-clip Python objects restated object for object since round 4 (ControlledPedestrian + P3dPose(nn.Module) + '
                      'one mock Transform / Location / Rotation per bone): 262 us per clip against 297 us for the reference\'s own '
                      'constructor timed in the build container, the whole step 513 ms there against 447.6 ms for the reference '
                      'in BASELINE.md section 2 (another, quieter 8-vCPU container)'}
    # one thread
    torch.set_num_threads(1)
    n1, t0 = 0, time.perf_counter()
    while True:
        P.port_train_step(model, opt, b['frames'], targets, meta)
        n1 += 1
        d1 = time.perf_counter() - t0
        if d1 >= seconds * 0.6 or n1 >= 50:
            break
    result['threads_1'] = {'value': round(batch_size * n1 / d1, 1), 'unit': 'clips/s', 'ms_per_step': round(d1 / n1 * 1e3, 1),
                           'sample': f'{n1} steps, 1 thread'}
    torch.set_num_threads(cores)
    result['gloo_ddp'] = cpu_ddp_baseline(cores, max(4.0, seconds * 0.5))
    return result


def timed_steps(trainer, flow, batches, steps, barrier):
    """EXACTLY `steps` train steps bracketed by barrier + synchronize; batches[i % len] is fed at step i."""
    n = len(batches)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = trainer.train_step(flow, batches[i % n], i)
    barrier()
    return time.perf_counter() - t0, loss


def extra_config(device, name, steps=100, warmup=10):
    """One more BASELINE.json configuration on this GPU: ms/step and clips/s of its captured train step (resident batch)."""
    from pedestrians_video_2_carla_amd.data.carla.carla_recorded_synthetic import SyntheticCarlaRecordedDataModule
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.trainer import Trainer, seed_everything
    seed_everything(22742)
    if name == 'cfg2':
        from pedestrians_video_2_carla_amd.modules.flow.pose_lifting import LitPoseLiftingFlow
        from pedestrians_video_2_carla_amd.modules.movements.linear_ae import LinearAE
        B = 1024
        model = LinearAE(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON)
        flow = LitPoseLiftingFlow(movements_model=model, loss_modes=['loc_2d_3d'], transform='hips_neck_bbox')
        workload = 'flow=pose_lifting LinearAE(pose_changes) loss=loc_2d_3d clip_length=16 batch_size=1024 (BASELINE.json configs[1])'
    elif name == 'cfg5':
        from pedestrians_video_2_carla_amd.modules.flow.pose_lifting import LitPoseLiftingFlow
        from pedestrians_video_2_carla_amd.modules.movements.pose_former import PoseFormer
        B, clip = 32, 81                      # configs[4]: 256 clips of 81 frames over 8 GPUs -> 32 per GPU
        model = PoseFormer(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, clip_length=clip)
        flow = LitPoseLiftingFlow(movements_model=model, loss_modes=['loc_2d_3d'], transform='hips_neck_bbox')
        workload = ('flow=pose_lifting PoseFormer(9 frames, 26 joints, E=32, depth 4, 8 heads; restated; K14 attention, K15 LayerNorm) '
                    'clip_length=81 absolute_loc head batch_size=32 = one GPU\'s share of BASELINE.json configs[4] (256 over 8 GPUs)')
    else:
        from pedestrians_video_2_carla_amd.modules.flow.autoencoder import LitAutoencoderFlow
        from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2SeqEmbeddings
        B = 512
        model = Seq2SeqEmbeddings(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_2d)
        flow = LitAutoencoderFlow(movements_model=model, loss_modes=['loc_2d'], transform='hips_neck_bbox')
        workload = ('flow=autoencoder Seq2SeqEmbeddings movements_output_type=pose_2d clip_length=16 batch_size=512 '
                    '(BASELINE.json configs[2])')
    dm = SyntheticCarlaRecordedDataModule(clip_length=81 if name == 'cfg5' else T_FRAMES, batch_size=B)
    trainer = Trainer(device=device, use_graph=True).setup(flow, dm)
    batch = dm.generate_batch(device)
    if name == 'cfg5':
        steps, warmup = 20, 3
    for i in range(warmup):
        trainer.train_step(flow, batch, i)
    torch.cuda.synchronize(device)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(steps):
            loss = trainer.train_step(flow, batch, i)
        torch.cuda.synchronize(device)
        times.append((time.perf_counter() - t0) / steps)
    dt = statistics.median(times)
    out = {'workload': workload, 'B': B, 'ms_per_step': round(dt * 1e3, 4), 'clips_per_s': round(B / dt, 1), 'dtype': 'f32',
           'hip_graph': True, 'steps': steps, 'final_loss': float(loss)}
    if name == 'cfg2':
        fused = fused_step_times(device, flow, trainer, batch)
        if fused is not None:                 # B = 1024 takes the two-launch step in its throughput form (a pair of wavefronts per clip)
            ft, fflops = fused
            out['step_breakdown_us'] = {k: round(v, 2) for k, v in ft.items()}
            out['roofline'] = [mfma_entry(k, B, ft[k], fl) for k, fl in fflops.items()]
        else:
            mt, flops = mlp_times(device, model, B, getattr(trainer, '_opt_in_backward', False))
            out['roofline'] = [mfma_entry(k, B, mt[k], fl) for k, fl in flops.items()]
        # configs[1] names bf16. The opt-in operand-precision arms of the fused MLP (LinearAE(mlp_precision='bf16' | 'bf16x3'), fp32
        # accumulate) are NOT timed here any more: measured in rounds 2-4 they are slower than this exact-fp32 step AND less accurate
        # (bf16: 0.089 ms at a 9 % loss deviation from the default init; split-bf16: 0.102 ms at 4.3e-5; fp32: this line) --
        # DESIGN.md section 8.4 keeps the record. dtype f32 is the reference's own arithmetic.
        out['note'] = ('dtype f32 = exact fp32 MFMA (bit-for-bit an fmaf chain), the reference\'s own arithmetic; the opt-in bf16 / '
                       'split-bf16 operand arms are slower than this step and less accurate (DESIGN.md section 8.4) and are no longer '
                       'part of the line')
    elif name == 'cfg5':
        # further arms of the same step: bf16 autocast on the GEMMs (K14 / K15 stay fp32), and the per-frame half of the
        # transformer run once per frame instead of once per (window, frame) -- in training that shares the stochastic-depth
        # drops of a frame between its windows (opt-in, PoseFormer(share_spatial=True)); fp32 per-window stays the headline
        def arm(**kw):
            try:
                seed_everything(22742)
                m = PoseFormer(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, clip_length=81, **kw)
                f = LitPoseLiftingFlow(movements_model=m, loss_modes=['loc_2d_3d'], transform='hips_neck_bbox')
                t = Trainer(device=device, use_graph=True).setup(f, dm)
                for i in range(3):
                    t.train_step(f, batch, i)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for i in range(steps):
                    la = t.train_step(f, batch, i)
                torch.cuda.synchronize(device)
                ms = (time.perf_counter() - t0) / steps * 1e3
                # (replayed_as_graph False = the trainer's capture-time check refused the captured step -- it says why on stderr -- and
                # the arm ran as eager launches: the bf16-autocast arms on this stack; the fp32 per-window headline is captured)
                return {'ms_per_step': round(ms, 3), 'clips_per_s': round(B / ms * 1e3, 1), 'final_loss': float(la),
                        'replayed_as_graph': bool(t.use_graph and getattr(t, '_graphs', None) is not None)}
            except Exception as e:                                      # noqa: BLE001
                return {'error': repr(e)[:200]}
        out['bf16_autocast'] = arm(compute_dtype=torch.bfloat16)
        out['shared_spatial'] = arm(share_spatial=True)
        out['shared_spatial_bf16'] = arm(share_spatial=True, compute_dtype=torch.bfloat16)
        out['note'] = ('transformer arithmetic parity-unpinned (third-party source absent); attention = K14 (p2c_attn_small), LayerNorm = '
                       'K15 (p2c_layernorm), every dense layer = K16 (p2c_gemm / p2c_gemm_tn: fp32 MFMA with bias / GELU / '
                       'stochastic-depth factor / residual in the epilogue, 110-130 TFLOP/s at the temporal shapes; no library GEMM in the step), narrow weight gradients = K12; '
                       'a block is one autograd node. The four temporal blocks are 2.8 TFLOP per step = 17.8 ms at the fp32 MFMA '
                       'peak; the pose head is the HIP absolute_loc kernel; stochastic depth (0.2) on')
        out['windows_per_step'] = B * (81 - 9 + 1)
        try:
            gt, gflops = gemm_times(device)
            out['roofline'] = [mfma_entry(k, B, gt[k], fl) for k, fl in gflops.items()]
        except Exception as e:                                      # noqa: BLE001 -- evidence beside the step time, not the step
            out['roofline'] = {'error': repr(e)[:200]}
    else:
        # the recurrences (K7b encoder layers, K7c decoder loop) priced against the fp32 MFMA peak: 2*T*B*H*4H flop per
        # LSTM layer and direction of the data flow (forward; the backward kernels do the same again with W^T)
        rt, rflops = recurrence_times(device, B)
        out['roofline'] = [mfma_entry(k, B, rt[k], fl) for k, fl in rflops.items()]
        out['note'] = ('the time-loop launches are latency chains (one to three workgroup barriers per time step); at B <= 4096 they '
                       'run 4 sequences per workgroup on v_mfma_f32_4x4x1_16B (128 workgroups at B = 512), above that 16 on '
                       'v_mfma_f32_16x16x4. The step is 19 launches: fold, 2 projection GEMMs + 1 input-gradient GEMM (K16: no '
                       'library GEMM in the step), 2 + 2 encoder recurrences, decoder fwd / bwd with the frame-invariant terms '
                       'inside, 2 grouped weight-gradient pairs, loss, AdamW; the inter-layer dropout masks are drawn inside the '
                       'time-loop kernels (no generator launch, no RNG-state fill in front of a replay)')
    return out


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, script=None, have=None):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks here, as the reference starts its own
    (modeling.py:275-282 -> Lightning re-runs the script once per GPU under `--gpus=0,1 --accelerator=ddp`, README.md:74-75).
    N fresh child processes of this script, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; this parent makes no GPU call
    (torch.cuda.device_count() only reads the device list) and does not replace itself. Rank 0's stdout is passed through
    (the ONE JSON line); any failing rank stops the others and the exit code is non-zero."""
    # (script / have: the rank program and the device count, for the CPU test of this launcher -- tests/test_host_logic.py)
    have = torch.cuda.device_count() if have is None else have
    if have < n:
        raise SystemExit(f'bench.py --gpus {n}: this host shows {have} GPU(s); refusing to report an n_gpus={n} line from fewer ranks')
    env0 = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(os.environ.get('MASTER_PORT') or _free_port()),
                WORLD_SIZE=str(n), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    procs = []
    try:
        for r in range(n):
            env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(script or __file__)] + list(argv), env=env,
                                          stdout=None if r == 0 else subprocess.DEVNULL))
        rc, pending = 0, list(procs)
        while pending and rc == 0:
            for pr in list(pending):
                code = pr.poll()
                if code is None:
                    continue
                pending.remove(pr)
                if code != 0:
                    rc = code
            time.sleep(0.05)
        return rc
    finally:
        for pr in procs:                                        # exactly the processes started here
            if pr.poll() is None:
                pr.terminate()
        for pr in procs:
            try:
                pr.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pr.kill()


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    from pedestrians_video_2_carla_amd.trainer import init_distributed
    info = init_distributed()
    world, rank, local_rank = info['world_size'], info['rank'], info['local_rank']
    if world != args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch as many ranks as GPUs (or none: '
                         f'`python bench.py --gpus N` starts them itself)')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP hot path has no CPU fallback')
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    # host side: a GPU box shows every core of the machine but gives a one-GPU job a share of 16. ATen's default of one intra-op
    # thread per visible core leaves hundreds of OpenMP workers spinning after each small CPU op (the synthetic batches are drawn
    # with a host generator) and can starve the launching thread of a host-bound loop (seen: fresh-batch step 0.07 -> 0.43 ms)
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16) // max(world, 1)))

    flow, dm, trainer, batch = build_step(device, args.batch_size, not args.no_graph, not args.full_outputs)
    for i in range(max(args.warmup, 1)):
        trainer.train_step(flow, batch, i)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    elapsed, loss = timed_steps(trainer, flow, [batch], args.steps, barrier)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss)

    # more timed blocks of the same K steps, and the same step on a NEW batch object every iteration (every rank runs them:
    # with an exchange in the step the collectives must pair up)
    repeats = [timed_steps(trainer, flow, [batch], args.steps, barrier)[0] / args.steps * 1e3 for _ in range(max(args.repeats, 0))]
    fresh = [dm.generate_batch(device, seed_offset=rank + 1000 * (k + 1)) for k in range(8)]
    timed_steps(trainer, flow, fresh, 16, barrier)
    fresh_ms = timed_steps(trainer, flow, fresh, args.steps, barrier)[0] / args.steps * 1e3
    allreduce_us = None
    if world > 1:
        # the collective alone, event-timed on the step's stream: what the step pays for the exchange on top of its kernels
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            trainer.exchange.all_reduce_gradients()
        e0.record()
        for _ in range(50):
            trainer.exchange.all_reduce_gradients()
        e1.record()
        e1.synchronize()
        allreduce_us = e0.elapsed_time(e1) * 1e3 / 50
        trainer.flat.zero_grad()

    if rank != 0:
        if world > 1:
            dist.barrier()                    # rank 0's last barrier; then out without RCCL's teardown (see the end of main)
            torch.cuda.synchronize()
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        return

    global_batch = args.batch_size * world
    value = global_batch * args.steps / elapsed
    cfg_name = {256: 'BASELINE.json metric config', 1024: 'BASELINE.json configs[1]' if world == 1 else
                'BASELINE.json configs[3]: 8192 global over 8 GPUs' if world == 8 else 'BASELINE.json configs[1]/[3] per-GPU batch'}
    result = {
        'metric': 'clips/sec (B=256,T=16,J=26) pose_lifting train step',
        'value': round(value, 1), 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'flow=pose_lifting movements_model=LinearAE(pose_changes) loss=loc_2d_3d '
                               f'transform=hips_neck_bbox clip_length={T_FRAMES} J={JOINTS} batch_size='
                               f'{args.batch_size}/GPU ({cfg_name.get(args.batch_size, "custom batch")}; configs[1]/[3] = '
                               f'--batch-size 1024)',
                   'global_batch': global_batch, 'per_gpu_batch': args.batch_size, 'parallelism': f'dp{world}',
                   'hip_graph': not args.no_graph, 'lean_train_outputs': not args.full_outputs,
                   'deferred_loss_finalize': os.environ.get('P2C_DEFER_FINALIZE', '2') != '0',
                   'two_launch_step': getattr(flow, '_pair_counts', None) is not None,
                   'direct_replay': getattr(trainer, '_direct', None) is not None,
                   'grad_allreduce_bytes': trainer.flat.nbytes(), 'final_loss': final_loss},
    }
    if repeats:
        result['repeat_ms_per_step'] = {'n': len(repeats), 'p50': round(statistics.median(repeats), 4),
                                        'min': round(min(repeats), 4), 'max': round(max(repeats), 4)}
    result['fresh_batch_ms_per_step'] = round(fresh_ms, 4)
    if allreduce_us is not None:
        result['allreduce_us'] = round(allreduce_us, 2)

    traffic = load_traffic()
    B = args.batch_size

    def with_traffic(entry):
        entry['traffic'] = (traffic.get(f'{entry["kernel"]}@B{B}') or {}).get('bytes')
        return entry

    entries, breakdown = {}, {}
    fused = fused_step_times(device, flow, trainer, batch) if world == 1 else None
    if fused is not None:
        ft, fflops = fused
        breakdown.update({k: round(v, 2) for k, v in ft.items()})
        for k, fl in fflops.items():
            entries[k] = with_traffic(mfma_entry(k, B, ft[k], fl))
        # the same launch against the HBM roofline: compulsory traffic of the step at the plugin boundary = frames + both
        # targets in (the model output and its gradient never exist in memory), nothing out but three scalars
        step_bytes = 4 * T_FRAMES * JOINTS * (2 + 2 + 3) + 4
        entries['train_clip_kernel']['hbm_view'] = roofline_entry('train_clip_kernel', B, ft['train_clip_kernel'], step_bytes)
    else:
        kt = kernel_times(device, B)
        names = head_kernel_names(B)
        per_clip = {'fwd': BYTES_FWD, 'bwd': BYTES_BWD}
        for w in ('fwd', 'bwd'):
            entries[names[w]] = with_traffic(roofline_entry(names[w], B, kt[w], per_clip[w]))
        if int(os.environ.get('P2C_DEFER_FINALIZE', '2')) == 2 and B <= 2048:
            breakdown['pose_head_train(count + fwd/bwd in one kernel + finalize)'] = round(kt['train'], 2)
        else:
            breakdown.update({names[w]: round(kt[w], 2) for w in ('fwd', 'bwd')})
        if getattr(flow.movements_model, 'fused_mlp', False):
            mt, flops = mlp_times(device, flow.movements_model, B, getattr(trainer, '_opt_in_backward', False))
            breakdown.update({k: round(v, 2) for k, v in mt.items()})
            for k, fl in flops.items():
                entries[k] = with_traffic(mfma_entry(k, B, mt[k], fl))
    result['step_breakdown_us'] = breakdown
    dominant = max(entries, key=lambda k: entries[k]['us_per_launch'])
    result['roofline'] = entries.pop(dominant)
    result['roofline']['other'] = list(entries.values())
    if not args.no_sweep and world == 1:
        per_clip = {'fwd': BYTES_FWD, 'bwd': BYTES_BWD}
        sweep = []
        for Bs in (256, 1024, 8192, 16384, 65536):
            k = kernel_times(device, Bs, reps=10 if Bs > 8192 else 20)
            names = head_kernel_names(Bs)
            for which in ('fwd', 'bwd'):
                e = roofline_entry(names[which], Bs, k[which], per_clip[which],
                                   (traffic.get(f'{names[which]}@B{Bs}') or {}).get('bytes'))
                if which == 'fwd':            # the op = this kernel + the one-workgroup loss reduction (a second launch)
                    e['us_per_op_with_loss_finalize'] = round(k['fwd_op'], 2)
                sweep.append(e)
        result['roofline_sweep'] = sweep
        # the FULL train step over batch sizes (SURVEY section 8d units: 1.61 MFLOP of LinearAE forward + dgrad + wgrad and 11 652 B of
        # frames + targets per clip): clips/s and both roofline fractions, with the launches the step consists of at that size
        try:
            result['step_sweep'] = step_sweep(device)
        except Exception as e:                                      # noqa: BLE001 -- a sweep point must not sink the headline
            result['step_sweep'] = {'error': repr(e)[:300]}
    if not args.no_extra_configs and world == 1:
        extra = {}
        for name in ('cfg2', 'cfg3', 'cfg5'):
            try:
                extra[name] = extra_config(device, name)
            except Exception as e:                                  # noqa: BLE001 -- an extra must not sink the headline
                extra[name] = {'error': repr(e)[:300]}
        result['extra_configs'] = extra
    if not args.no_cpu_baseline and world == 1:
        result['cpu_baseline'] = cpu_baseline(args.batch_size, args.cpu_seconds)
    print(json.dumps(result), flush=True)
    if world > 1:
        # every rank has finished; leave WITHOUT tearing the communicator down: destroy_process_group with a captured graph that
        # holds the collective aborted the interpreter once in a while inside RCCL's teardown (tests/test_ddp_gpu.py), and a rank
        # that dies after the line is printed would still fail the launcher
        dist.barrier()
        torch.cuda.synchronize()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
