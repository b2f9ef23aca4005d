"""Developer tool: phase timeline of the two-launch train step (needs a build with EXTRA=-DP2C_TRAIN_TRACE)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from test_flow_gpu import make, dev
from pedestrians_video_2_carla_amd import _lib
from pedestrians_video_2_carla_amd.trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
flow, dm = make(B=B, missing=0.0)
tr = Trainer(device=dev(), use_graph=True).setup(flow, dm)
batch = dm.generate_batch(dev())
for i in range(30):
    tr.train_step(flow, batch, i)
torch.cuda.synchronize()
lib = _lib.lib()
lib.p2c_debug_train_trace.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 80)()
assert lib.p2c_debug_train_trace(buf) == 0
for k, name in ((0, 'train_clip'), (1, 'train_wgrad')):
    t = list(buf[40 * k:40 * k + 40])
    cyc, wall = t[39] - t[0], (t[37] - t[38]) * 10.0       # wall clock: 100 MHz -> ns
    print(f'B={B} {name}: {cyc} cycles, {wall:.0f} ns, {cyc / max(wall, 1):.2f} GHz')
    order = sorted((v, i) for i, v in enumerate(t[:37]) if v >= t[0] and i > 0) + [(t[39], 39)]
    prev = t[0]
    for v, i in order:
        print(f'   [{i:2d}] +{v - prev:6d}')
        prev = v
