"""Train-step time of the pose-lifting flow (LinearAE, loc_2d_3d, HIP graph) at several batch sizes, two-launch step on / off."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

out = {}
for B in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096]:
    for fused in ('1', '0'):
        os.environ['P2C_FUSED_TRAIN'] = fused
        os.environ['P2C_FUSED_TRAIN_MAX_B'] = str(1 << 20)
        flow, dm, trainer, batch = bench.build_step(torch.device('cuda:0'), B, True, True)
        for i in range(20):
            trainer.train_step(flow, batch, i)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(200):
                trainer.train_step(flow, batch, i)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 200 * 1e6)
        out[f'B{B}_fused{fused}'] = round(min(ts), 1)
        print(B, fused, round(min(ts), 1), 'us', flush=True)
print(json.dumps(out))
