"""Developer tool: phase timeline of one workgroup of K16's NT kernel (build with EXTRA=-DP2C_GEMM_TRACE, P2C_LIB_PATH=that build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import _lib, ops
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (21024, 2496, 832)
d = torch.device('cuda:0')
a, w = torch.randn(M, K, device=d), torch.randn(N, K, device=d)
for _ in range(int(os.environ.get('GEMMTRACE_WARM', '60'))):       # long enough for the clocks to settle under load
    ops.gemm(a, w, True)
torch.cuda.synchronize()
lib = _lib.lib()
lib.p2c_debug_gemm_trace.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 64)()
assert lib.p2c_debug_gemm_trace(buf) == 0
t = list(buf)
print('first fetch issue', t[1] - t[0], '| wait + commit', t[2] - t[1], '| barrier', t[3] - t[2])
for kt in range(8):
    b = 4 + kt * 6
    print(f'k-tile {kt}: MFMA phase {t[b + 1] - t[b]:6d} | barrier {t[b + 2] - t[b + 1]:5d} | vmcnt wait {t[b + 3] - t[b + 2]:5d} | '
          f'LDS stores {t[b + 4] - t[b + 3]:5d} | barrier {t[b + 5] - t[b + 4]:5d} | next fetch issue {t[b + 6] - t[b + 5] if kt < 7 else 0:5d}')
print('main loop', t[60] - t[3], 'epilogue', t[61] - t[60], 'workgroup total', t[61] - t[0])
print('shader clock over this workgroup\'s life: %.0f MHz (%.1f us)' % ((t[61] - t[0]) / ((t[63] - t[62]) / 100.0), (t[63] - t[62]) / 100.0))
