"""Host time of enqueueing one frame (hipexec_render_device returns before the GPU has finished) against the frame's
GPU time, for rank 0's share at G = 1, 2, 4, 8: python tools/submit_cost.py c2"""
import os, sys, time
import os as _os; _os.environ.setdefault("HIPEXEC_KNOBS", "1")   # the HIPEXEC_* experiment knobs are honoured only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import scenegen
from raytracer_amd.executor import FramesInFlight
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
W, H, depth, fn = {"c2": (1920, 1080, 3, scenegen.scene_c2), "c3": (3840, 2160, 5, scenegen.scene_c2)}[wl]
fs = fn().to_flat()
K = 2
pipe = FramesInFlight(fs, device=0, depth=K)
outs = [torch.zeros((((H + 7) // 8) * 8, W, 4), dtype=torch.float32, device="cuda") for _ in range(K)]
for G in (1, 2, 4, 8):
    def one():
        k = pipe.next
        t0 = time.perf_counter()
        pipe.submit(W, H, outs[k].data_ptr(), None, max_depth=depth, tile_first=0, tile_step=G)
        return time.perf_counter() - t0
    for _ in range(6):
        one()
    pipe.drain()
    # host cost with an idle GPU behind it: submit, then wait, one frame at a time
    host = []
    for _ in range(20):
        host.append(one()); pipe.drain()
    # throughput with frames in flight
    t0 = time.perf_counter()
    sub = [one() for _ in range(40)]
    pipe.drain()
    per = (time.perf_counter() - t0) / 40
    host.sort(); sub.sort()
    print(f"{wl} G={G}: submit call {host[len(host)//2]*1e3:.3f} ms (median, idle GPU), {sub[len(sub)//2]*1e3:.3f} ms (median, in flight; includes "
          f"waiting for the slot's previous frame), frame period {per*1e3:.3f} ms, launches {sum(pipe.stats[0].launches)}", flush=True)
pipe.close()
