"""Throughput with several frames in flight: K executors on the same scene take the frames in turn, each frame
asynchronous and device-resident; a frame is collected just before its executor is needed again.
python tools/pipeline_times.py c2 [G]   (G: rank 0's share of a frame split over G ranks)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import scenegen, _abi
from raytracer_amd.executor import HipExecutor
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H, depth, fn = {"c2": (1920, 1080, 3, scenegen.scene_c2), "c3": (3840, 2160, 5, scenegen.scene_c2),
                   "c5": (3840, 2160, 0, scenegen.scene_c5), "c1": (640, 480, 0, scenegen.scene_c1),
                   "c4": (1920, 1080, 0, scenegen.scene_c4)}[wl]
fs = fn().to_flat()
for K in (1, 2, 3):
    exs = [HipExecutor(fs, device=0) for _ in range(K)]
    outs = [torch.zeros((((H + 7) // 8) * 8, W, 4), dtype=torch.float32, device="cuda") for _ in range(K)]
    sts = [_abi.Stats() for _ in range(K)]
    busy = [False] * K
    def frame(i):
        k = i % K
        if busy[k]:
            exs[k].sync()
        exs[k].render_device(W, H, outs[k].data_ptr(), None, max_depth=depth, tile_first=0, tile_step=G, stats=sts[k])
        busy[k] = True
    def drain():
        for k in range(K):
            if busy[k]:
                exs[k].sync(); busy[k] = False
    for i in range(3 * K):
        frame(i)
    drain()
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for i in range(N):
        frame(i)
    drain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e3
    print(f"{wl} G={G} frames in flight {K}: {dt:.3f} ms per frame, {sts[0].rays / dt / 1e3:.0f} Mrays/s", flush=True)
    if K > 1:
        assert torch.equal(outs[0], outs[1])
    for e in exs:
        e.close()
