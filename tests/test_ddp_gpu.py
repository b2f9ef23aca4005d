"""The multi-GPU step on the one-GPU box: RCCL process group of ONE rank with the gradient exchange forced on, so the
two-graph step (graph A: forward + backward -> eager RCCL all-reduce of the flat gradient buffer -> graph B: FlatAdamW with
the 1/world factor folded in) runs exactly as it does with 2..8 ranks. With one rank the all-reduce is the identity, so the
loss curve must match the single-graph trainer's.

Every case runs in a CHILD process (this file executed as a script) that prints its verdict and leaves through ``os._exit``
without tearing the communicator down: ``destroy_process_group`` with a captured graph that holds the collective aborted the
interpreter once in a while on the GPU box (inside RCCL's teardown, after the case had passed), and an abort of the test
runner takes every later test with it. The assertions are unchanged; a failing case prints its traceback and the parent test
fails with it."""
import os
import subprocess
import sys
import traceback

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OK = 'P2C_DDP_CASE_OK'


def _run_child(case, *args, port):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), P2C_FORCE_EXCHANGE='1', PYTHONPATH=os.pathsep.join(
        [ROOT, os.path.join(ROOT, 'tests')] + [p for p in os.environ.get('PYTHONPATH', '').split(os.pathsep) if p]))
    res = subprocess.run([sys.executable, os.path.abspath(__file__), case, *args], env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0 and OK in res.stdout, f'{case}{args}: rc={res.returncode}\n{res.stdout[-4000:]}\n{res.stderr[-4000:]}'


@pytest.mark.parametrize('captured', ['auto', '0'])
def test_distributed_step_with_rccl_matches_single_graph_step(captured):
    """captured = 'auto': the all-reduce is part of the one step graph (default; guarded by a first-replay check with an eager
    fall-back); '0': two stages with the eager collective between them (P2C_GRAPH_ALLREDUCE=0)."""
    _run_child('rccl_step', captured, port=29533)


def test_failed_capture_of_the_collective_falls_back_to_the_eager_path():
    """If capturing (or first replaying) the all-reduce fails on any rank, every rank agrees on the eager-collective step and
    training goes on with unchanged state."""
    _run_child('failed_capture', port=29534)


@pytest.mark.parametrize('captured', ['auto', '0'])
def test_cfg4_per_gpu_step_with_the_exchange_on(captured):
    """BASELINE.json configs[3] per GPU: B = 1024 clips, the two-launch step WITHOUT the optimizer in its second launch
    (train_clip_kernel -> train_wgrad_kernel<false> -> RCCL all-reduce of the flat gradient -> p2c_adamw_step), captured as one
    graph ('auto') and as two stages with the eager collective between ('0'), against the single-GPU two-launch step on the
    same batch. The captured step's graph is counted through the C ABI: the three kernels plus whatever the collective adds
    (a one-rank in-place all-reduce may add a kernel node or nothing; a memcpy / memset node would be a regression)."""
    _run_child('cfg4_step', captured, port=29535)


# ---- the cases (child process) -------------------------------------------------------------------------------------------
def _case_rccl_step(captured):
    import torch
    import torch.distributed as dist
    from test_flow_gpu import make
    from pedestrians_video_2_carla_amd.trainer import Trainer
    assert torch.cuda.is_available(), 'needs the MI355X'
    d = torch.device('cuda:0')
    torch.cuda.set_device(d)
    steps = 12
    flow_a, dm = make(B=16, missing=0.0)
    flow_b, _ = make(B=16, missing=0.0)
    batch = dm.generate_batch(d)
    os.environ.pop('P2C_FORCE_EXCHANGE', None)
    ta = Trainer(device=d, use_graph=True).setup(flow_a, dm)
    single = torch.stack([ta.train_step(flow_a, batch, i).clone() for i in range(steps)]).cpu()

    os.environ['P2C_FORCE_EXCHANGE'] = '1'
    os.environ['P2C_GRAPH_ALLREDUCE'] = captured
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    tb = Trainer(device=d, use_graph=True).setup(flow_b, dm)
    assert tb.exchange.enabled and tb.exchange.world == 1 and not tb.exchange.average_here
    multi = torch.stack([tb.train_step(flow_b, batch, i).clone() for i in range(steps)]).cpu()
    if captured == '0':
        assert tb._graphs[1] is not None, 'two stages with the eager collective between them'
    else:
        assert tb._graphs[1] is None, 'the collective is part of the one step graph'
    dist.barrier()
    # the single-GPU trainer applies AdamW inside the MLP backward's reduction, the multi-GPU one as its own launch after
    # the all-reduce: same formula, not the same instruction stream -> equal to fp32 rounding, amplified by the default
    # initialisation, not bit for bit
    assert torch.allclose(single, multi, rtol=2e-4, atol=0), (single, multi)
    pa = torch.cat([p.detach().reshape(-1) for p in flow_a.parameters()])
    pb = torch.cat([p.detach().reshape(-1) for p in flow_b.parameters()])
    # Adam moves a parameter by ~lr (1e-4) per step whatever the size of its gradient: where the gradient is rounding
    # noise the two instruction streams may step differently -- a few lr over the 12 steps at most
    assert float((pa - pb).abs().max()) <= 3e-4


def _case_failed_capture():
    import contextlib
    import io
    import torch
    import torch.distributed as dist
    from test_flow_gpu import make
    from pedestrians_video_2_carla_amd.parallel.flat import GradientExchange
    from pedestrians_video_2_carla_amd.trainer import Trainer
    d = torch.device('cuda:0')
    torch.cuda.set_device(d)
    flow_a, dm = make(B=16, missing=0.0)
    flow_b, _ = make(B=16, missing=0.0)
    batch = dm.generate_batch(d)
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    os.environ['P2C_GRAPH_ALLREDUCE'] = '0'
    ta = Trainer(device=d, use_graph=True).setup(flow_a, dm)
    want = torch.stack([ta.train_step(flow_a, batch, i).clone() for i in range(6)]).cpu()

    os.environ['P2C_GRAPH_ALLREDUCE'] = 'auto'
    real = GradientExchange.all_reduce_gradients

    def broken(self):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('collective not capturable (simulated)')
        return real(self)
    GradientExchange.all_reduce_gradients = broken
    printed = io.StringIO()
    with contextlib.redirect_stderr(printed):             # (the trainer's notices go to stderr: stdout is bench.py's one JSON line)
        tb = Trainer(device=d, use_graph=True).setup(flow_b, dm)
        got = torch.stack([tb.train_step(flow_b, batch, i).clone() for i in range(6)]).cpu()
    assert tb._graphs[1] is not None                       # the eager-collective structure
    assert 'using the eager collective' in printed.getvalue(), printed.getvalue()
    assert torch.equal(want, got), (want, got)


def _case_cfg4_step(captured):
    import torch
    import torch.distributed as dist
    from test_flow_gpu import make
    from pedestrians_video_2_carla_amd.trainer import Trainer
    assert torch.cuda.is_available(), 'needs the MI355X'
    d = torch.device('cuda:0')
    torch.cuda.set_device(d)
    steps, B = 8, 1024
    flow_a, dm = make(B=B, missing=0.1)
    flow_b, _ = make(B=B, missing=0.1)
    batch = dm.generate_batch(d)
    os.environ.pop('P2C_FORCE_EXCHANGE', None)
    ta = Trainer(device=d, use_graph=True).setup(flow_a, dm)
    single = torch.stack([ta.train_step(flow_a, batch, i).clone() for i in range(steps)]).cpu()
    assert ta._direct is not None, 'B = 1024 takes the two-launch step on one GPU'

    os.environ['P2C_FORCE_EXCHANGE'] = '1'
    os.environ['P2C_GRAPH_ALLREDUCE'] = captured
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    tb = Trainer(device=d, use_graph=True).setup(flow_b, dm)
    assert tb.exchange.enabled and not tb._opt_in_backward
    multi = torch.stack([tb.train_step(flow_b, batch, i).clone() for i in range(steps)]).cpu()
    assert getattr(flow_b, '_pair_counts', None) is not None, 'the exchange step is the two-launch step too'
    nodes = tb._graph_nodes
    assert nodes is not None, 'the captured graph could not be counted'
    total, kernels = nodes
    if captured == '0':
        assert tb._graphs[1] is not None and (total, kernels) == (2, 2), nodes      # stage A: clip + wgrad<false>
    else:
        assert tb._graphs[1] is None and total == kernels and kernels in (3, 4), nodes
    dist.barrier()
    assert torch.allclose(single, multi, rtol=2e-4, atol=0), (single, multi)
    pa = torch.cat([p.detach().reshape(-1) for p in flow_a.parameters()])
    pb = torch.cat([p.detach().reshape(-1) for p in flow_b.parameters()])
    assert float((pa - pb).abs().max()) <= 3e-4


if __name__ == '__main__':
    code = 1
    try:
        {'rccl_step': _case_rccl_step, 'failed_capture': _case_failed_capture, 'cfg4_step': _case_cfg4_step}[sys.argv[1]](*sys.argv[2:])
        import torch
        torch.cuda.synchronize()
        print(OK, flush=True)
        code = 0
    except BaseException:                                   # noqa: BLE001 -- the verdict has to reach the parent
        traceback.print_exc()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)                                          # (no interpreter / RCCL teardown: see the module docstring)
