import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
torch.set_num_threads(16)
d = torch.device('cuda:0')
flow, dm, trainer, batch = bench.build_step(d, 256, True, True)
for i in range(5):
    trainer.train_step(flow, batch, i)
fresh = [dm.generate_batch(d, seed_offset=1000 * (k + 1)) for k in range(8)]
for i in range(32):
    trainer.train_step(flow, fresh[i % 8], i)
torch.cuda.synchronize()
N = 400
def T(f):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N): f(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6
print('train_step fresh     ', T(lambda i: trainer.train_step(flow, fresh[i % 8], i)))
print('train_step resident  ', T(lambda i: trainer.train_step(flow, fresh[0], i)))
print('stage_batch only     ', T(lambda i: trainer.stage_batch(flow, fresh[i % 8], i)))
g_fb, g_opt = trainer._graphs
print('replay only          ', T(lambda i: g_fb.replay()))
print('with_skel_type       ', T(lambda i: trainer._with_skel_type(fresh[i % 8])))
frames, targets, meta = fresh[1]
src = [frames if k == 'frames' else (targets[k[8:]] if k[0] == 't' else meta[k[5:]]) for k in trainer._static_names]
print('n tensors', len(src), trainer._static_names)
print('foreach_copy         ', T(lambda i: torch._foreach_copy_(trainer._static_dst, src)))
print('batch_start hook     ', T(lambda i: flow.on_train_batch_start(trainer._static_batch, i)))
def lists(i):
    frames, targets, meta = fresh[i % 8]
    s = [frames if k == 'frames' else (targets[k[8:]] if k[0] == 't' else meta[k[5:]]) for k in trainer._static_names]
    n = 1 + sum(isinstance(v, torch.Tensor) for v in targets.values()) + sum(isinstance(v, torch.Tensor) for v in meta.values())
    ok = n != len(s) or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(s, trainer._static_dst))
print('list + checks        ', T(lists))
for o in trainer.optimizers:
    if hasattr(o, 'sync_hyper'):
        print('sync_hyper           ', T(lambda i: o.sync_hyper()))
sb = trainer._static_batch
print('  projection hook    ', T(lambda i: flow.projection.on_batch_start(sb, i)))
print('  plan lookup        ', T(lambda i: flow._fused_train_plan(sb[0], sb[1])))
pc = getattr(flow, '_pair_counter', None)
if pc is not None:
    gt = flow._fused_train_plan(sb[0], sb[1])[1]
    print('  pair counter call  ', T(lambda i: pc(gt)))
d = trainer._direct
if d is not None:
    print('direct call only     ', T(lambda i: d['call'](d['desc_ref'], d['gl'], torch.cuda.current_stream().cuda_stream)))
    print('  current_stream     ', T(lambda i: torch.cuda.current_stream().cuda_stream))
