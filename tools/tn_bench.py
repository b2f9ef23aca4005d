import torch, sys
sys.path.insert(0,'.')
from pedestrians_video_2_carla_amd import ops
d=torch.device('cuda:0')
for K,M,N in [(21024,2496,832),(21024,832,832),(21024,1664,832),(21024,832,1664)]:
    a=torch.randn(K,M,device=d); b=torch.randn(K,N,device=d)
    for f,name in ((lambda: ops.gemm_tn(a,b),'tn'),(lambda: torch.mm(a.t(),b),'lib')):
        for _ in range(3): f()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/10
        print(K,M,N,name,'%.1f us %.1f TF'%(t*1e3, 2*K*M*N/t/1e9))
