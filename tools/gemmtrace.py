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
buf = (ctypes.c_ulonglong * 65)()
assert lib.p2c_debug_gemm_trace(buf) == 0
t = list(buf)
print('first fetch issue', t[1] - t[0], '| wait + commit', t[2] - t[1], '| barrier', t[3] - t[2])
for kt in range(8):
    b = 4 + kt * 6
    print(f'k-tile {kt}: MFMA phase {t[b + 1] - t[b]:6d} | barrier {t[b + 2] - t[b + 1]:5d} | vmcnt wait {t[b + 3] - t[b + 2]:5d} | '
          f'LDS stores {t[b + 4] - t[b + 3]:5d} | barrier {t[b + 5] - t[b + 4]:5d} | next fetch issue {t[b + 6] - t[b + 5] if kt < 7 else 0:5d}')
print('main loop', t[60] - t[3], 'epilogue', t[61] - t[60], 'workgroup total', t[61] - t[0])
print('shader clock over this workgroup\'s life: %.0f MHz (%.1f us)' % ((t[61] - t[0]) / ((t[63] - t[62]) / 100.0), (t[63] - t[62]) / 100.0))

# the same launch seen from workgroups that start in later rounds: life in cycles, wall-clock start / end relative to the first
lib.p2c_debug_gemm_trace_block.argtypes = [ctypes.c_int]
rows = []
for blk in (0, 640, 1279, 1280, 2000, 3000, 4000, 5000, 6000, 6300, 6434):
    assert lib.p2c_debug_gemm_trace_block(blk) == 0
    ops.gemm(a, w, True)
    torch.cuda.synchronize()
    assert lib.p2c_debug_gemm_trace(buf) == 0
    t = list(buf)
    rows.append((blk, t[61] - t[0], t[3] - t[0], t[60] - t[3], t[61] - t[60], (t[62] - t[64]) / 100.0, (t[63] - t[64]) / 100.0))
for blk, life, pro, main, epi, w0, w1 in rows:
    print(f'block {blk:5d}: life {life:7d} cycles (prologue {pro:6d}, main {main:7d}, epilogue {epi:6d}), from {w0:7.1f} to {w1:7.1f} us after workgroup 0 started')
