"""Developer tool: phase timeline of K14's backward for one sequence (needs a build with EXTRA=-DP2C_ATTN_TRACE)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import _lib, ops
S, N, Hh, D = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (2336, 9, 8, 104)
d = torch.device('cuda:0')
qkv = torch.randn(S, N, 3, Hh, D, device=d, requires_grad=True)
for _ in range(3):
    out = ops.small_attention(qkv, D ** -0.5)
    out.sum().backward()
torch.cuda.synchronize()
lib = _lib.lib()
lib.p2c_debug_attn_trace.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 16)()
assert lib.p2c_debug_attn_trace(buf) == 0
t = list(buf)
names = ['copy-in', 'scores + dP', 'softmax + dS', 'gradients out']
for i, n in enumerate(names):
    print(f'{n:16s} {t[i + 1] - t[i]:8d} cycles')
print('sequence total  ', t[4] - t[0])
if t[5]:
    print('  dQ', t[5] - t[3], 'dK', t[6] - t[5], 'dV', t[7] - t[6], 'drain + barrier', t[4] - t[7])
