"""Throughput of rank 0's share (G = 1, 2, 4, 8) with frames in flight (executors of one scene, one lane each, taking
frames in turn) against one executor with two lanes: python tools/pipeline_times.py c2 [depths of flight ...]"""
import os, sys, time
import os as _os; _os.environ.setdefault("HIPEXEC_KNOBS", "1")   # the HIPEXEC_* experiment knobs are honoured only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import scenegen
from raytracer_amd.executor import FramesInFlight
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
Ks = [int(a) for a in sys.argv[2:]] or [1, 2]
W, H, depth, fn = {"c2": (1920, 1080, 3, scenegen.scene_c2), "c3": (3840, 2160, 5, scenegen.scene_c2),
                   "c5": (3840, 2160, 0, scenegen.scene_c5), "c1": (640, 480, 0, scenegen.scene_c1),
                   "c4": (1920, 1080, 0, scenegen.scene_c4)}[wl]
fs = fn().to_flat()
for K in Ks:
    pipe = FramesInFlight(fs, device=0, depth=K)
    outs = [torch.zeros((((H + 7) // 8) * 8, W, 4), dtype=torch.float32, device="cuda") for _ in range(K)]
    base = None
    for G in (1, 2, 4, 8):
        def run(n):
            for _ in range(n):
                k = pipe.next
                pipe.submit(W, H, outs[k].data_ptr(), None, max_depth=depth, tile_first=0, tile_step=G)
            pipe.drain()
        run(3 * K)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            N = 20
            t0 = time.perf_counter(); run(N); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / N * 1e3)
        base = base or best
        print(f"{wl} in flight {K} G={G}: {best:.3f} ms per frame, {pipe.stats[0].rays / best / 1e3:.0f} Mrays/s, ideal {base / G:.3f}, "
              f"efficiency {base / G / best:.2f}", flush=True)
    pipe.close()
