"""Time the per-frame executor pattern of ray_tracer_cl.cpp:14 (create, render one frame, destroy)."""
import os, sys, time
import os as _os; _os.environ.setdefault("HIPEXEC_KNOBS", "1")   # the HIPEXEC_* experiment knobs are honoured only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import scenegen, _abi
from raytracer_amd.executor import HipExecutor
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
fs = {"c2": scenegen.scene_c2, "c5": scenegen.scene_c5, "c4": scenegen.scene_c4, "c1": scenegen.scene_c1}[wl]().to_flat()
W, H = 800, 600
out = np.zeros((H, W, 3), np.float32)
os.environ["HIPEXEC_TRACE_CREATE"] = os.environ.get("HIPEXEC_TRACE_CREATE", "0")
for i in range(5):
    t0 = time.perf_counter()
    ex = HipExecutor(fs, device=0)
    t1 = time.perf_counter()
    ex.render(W, H, mode=_abi.HIPEXEC_MODE_CL, want_ids=False, out=out)
    t2 = time.perf_counter()
    ex.close()
    t3 = time.perf_counter()
    print(wl, f"create {1e3*(t1-t0):.2f} ms, first frame {1e3*(t2-t1):.2f} ms, destroy {1e3*(t3-t2):.2f} ms, total {1e3*(t3-t0):.2f} ms", flush=True)
