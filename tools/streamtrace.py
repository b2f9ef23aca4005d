"""Developer tool: phase timeline of train_stream_kernel, wave 0 of one workgroup, LAST clip it walked
(needs a build with EXTRA=-DP2C_STREAM_TRACE; P2C_LIB_PATH selects it)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from test_flow_gpu import make, dev
from pedestrians_video_2_carla_amd import _lib
from pedestrians_video_2_carla_amd.trainer import Trainer

os.environ['P2C_FUSED_TRAIN_MAX_B'] = str(1 << 20)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = _lib.lib()
lib.p2c_train_step_set_stream_min_batch(1)
flow, dm = make(B=B, missing=0.1)
tr = Trainer(device=dev(), use_graph=True).setup(flow, dm)
batch = dm.generate_batch(dev())
for i in range(30):
    tr.train_step(flow, batch, i)
torch.cuda.synchronize()
lib.p2c_debug_stream_trace.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 64)()
assert lib.p2c_debug_stream_trace(buf) == 0
t = list(buf)
names = {1: 'prologue', 2: 'loop top', 3: 'target loads issued', 4: 'fwd L0 + store + sync', 5: 'fwd L1', 6: 'fwd L2', 7: 'fwd L3', 8: 'fwd L4',
         9: 'fwd L5', 20: 'sync (y complete)', 21: 'read y, 6D -> R, tables', 22: 'prefix scan', 23: 'hand-over 1', 24: 'R, FK',
         25: 'head4', 26: 'subtree, torque', 27: 'suffix scan, loss sums', 28: 'hand-over 2 + pull-back + write', 32: 'sync (grad_y complete)',
         33: 'G6 store + dgrad L5 + sync', 34: 'dgrad L4 + x commit', 35: 'dgrad L3', 36: 'dgrad L2', 37: 'dgrad L1 + G1 store', 43: 'change[0] x reference', 44: 'scan round 0', 45: 'scan round 1', 63: 'sync (end)'}
cyc, wall = t[63] - t[0], (t[61] - t[62]) * 10.0
print(f'B={B}: kernel (wave 0) {cyc} cycles, {wall:.0f} ns, {cyc / max(wall, 1):.2f} GHz; last clip {t[63] - t[2]} cycles')
order = sorted((v, i) for i, v in enumerate(t[:61]) if v >= t[2] and i > 2 and (i < 40 or 43 <= i <= 50)) + [(t[63], 63)]
prev = t[2]
for v, i in order:
    print(f'   [{i:2d}] +{v - prev:6d}  {names.get(i, "")}')
    prev = v
print(f'   prologue {t[1] - t[0]} cycles: loads issued +{t[40] - t[0]}, landed + barrier +{t[41] - t[40]}, coefficients +{t[42] - t[41]}, late pieces issued +{t[1] - t[42]}')
