"""A few launches of K16 at cfg5's big shapes (for rocprofv3 --pmc passes):  python tools/gemm_pmc.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import ops
d = torch.device('cuda:0')
M, N, K = 21024, 2496, 832
a, w = torch.randn(M, K, device=d), torch.randn(N, K, device=d)
gy = torch.randn(M, N, device=d)
for _ in range(4):
    ops.gemm(a, w, True)          # NT
    ops.gemm(gy, w, False)        # NN
    ops.gemm_tn(gy, a)            # TN
    torch.mm(a, w.t())            # library, for comparison
torch.cuda.synchronize()
print('done')
