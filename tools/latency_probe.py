import time, numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
import yolov3_tensorflow_amd as y3, bench
for dt in ('f32', 'f32_bf16x6'):
    y3.reset_default_graph()
    model = y3.yolov3(80, bench.ANCHORS); model.compute_dtype = dt
    for bs in (1, 4, 8):
        x = torch.rand((bs, 416, 416, 3), device='cuda')
        with y3.variable_scope('yolov3'):
            if bs == 1:
                model.forward(torch.zeros((1, 64, 64, 3), device='cuda')); bench.random_init(1)
            for _ in range(5): model.forward(x)
            lat = []
            for _ in range(30):
                torch.cuda.synchronize(); t = time.perf_counter(); model.forward(x); torch.cuda.synchronize()
                lat.append(time.perf_counter() - t)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(30): model.forward(x)
            torch.cuda.synchronize(); thr = (time.perf_counter() - t) / 30
        print(dt, 'bs', bs, 'p50 latency %.3f ms' % (np.median(lat) * 1e3), 'pipelined %.3f ms/step' % (thr * 1e3))
