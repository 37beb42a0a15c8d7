"""Wall time of hipexec_render (host buffers, PCIe included) against the device time of the same frame."""
import os, sys, time
import os as _os; _os.environ.setdefault("HIPEXEC_KNOBS", "1")   # the HIPEXEC_* experiment knobs are honoured only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import scenegen
from raytracer_amd.executor import HipExecutor
ex = HipExecutor(scenegen.scene_c2().to_flat(), device=0)
W, H = 1920, 1080
out = np.zeros((H, W, 3), np.float32)
ts = []
for i in range(8):
    t0 = time.perf_counter()
    r = ex.render(W, H, max_depth=3, want_ids=False, out=out)
    ts.append((time.perf_counter() - t0) * 1e3)
print("hipexec_render C2: wall min %.2f ms median %.2f ms, device %.2f ms" % (min(ts[2:]), sorted(ts[2:])[3], r["stats"].total_ms))
