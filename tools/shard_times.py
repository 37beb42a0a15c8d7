"""Time one rank's share of a frame for G = 1, 2, 4, 8 (tile_first = 0, tile_step = G) on one GPU:
the compute part of the strong-scaling curve without the gather."""
import os, sys, time
import os as _os; _os.environ.setdefault("HIPEXEC_KNOBS", "1")   # the HIPEXEC_* experiment knobs are honoured only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import scenegen, _abi
from raytracer_amd.executor import HipExecutor
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
W, H, depth, fn = {"c2": (1920, 1080, 3, scenegen.scene_c2), "c3": (3840, 2160, 5, scenegen.scene_c2),
                   "c5": (3840, 2160, 0, scenegen.scene_c5), "c1": (640, 480, 0, scenegen.scene_c1),
                   "c4": (1920, 1080, 0, scenegen.scene_c4)}[wl]
ex = HipExecutor(fn().to_flat(), device=0)
out = torch.zeros((((H + 7) // 8) * 8, W, 4), dtype=torch.float32, device="cuda")
base = None
for G in (1, 2, 4, 8):
    st = _abi.Stats()
    ts = []
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ex.render_device(W, H, out.data_ptr(), None, max_depth=depth, tile_first=0, tile_step=G, stats=st)
        ex.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    t = min(ts[2:])
    base = base or t
    print(f"{wl} G={G}: rank-0 share {t:.3f} ms wall ({st.total_ms:.3f} ms device, lanes {st.lanes}), ideal {base / G:.3f} ms, "
          f"compute efficiency {base / G / t:.2f}", flush=True)
