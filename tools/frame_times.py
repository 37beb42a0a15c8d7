"""Print per-frame kernel-class times of a workload (diagnostic)."""
import sys, os
import os as _os; _os.environ.setdefault("HIPEXEC_KNOBS", "1")   # the HIPEXEC_* experiment knobs are honoured only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scenegen
from raytracer_amd.executor import HipExecutor
import numpy as np
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H, depth, fs = {"c2": (1920, 1080, 3, scenegen.scene_c2), "c5": (3840, 2160, 0, scenegen.scene_c5),
                   "c1": (640, 480, 0, scenegen.scene_c1), "c3": (3840, 2160, 5, scenegen.scene_c2), "c4": (1920, 1080, 0, scenegen.scene_c4)}[wl]
fs = fs().to_flat()
ex = HipExecutor(fs, device=0)
out = np.zeros((H, W, 3), np.float32)
for i in range(n):
    r = ex.render(W, H, max_depth=depth, want_ids=False, time_kernels=True, out=out, algo=os.environ.get("ALGO"),
                  tile_step=int(os.environ.get("TILE_STEP", "1")))
    st = r["stats"]
    print(i, "total_ms", round(st.total_ms, 1), [round(x, 1) for x in st.kernel_ms], list(st.launches), "rays", st.rays, "exact_groups", list(st.exact_groups), "scanned_chunks", list(st.scanned_chunks),
          "rays p/b/s", st.rays_primary, st.rays_bounce, st.rays_shadow, "bvh_nodes", list(st.bvh_nodes), "bvh_tests", list(st.bvh_tests), "ovf", st.bvh_stack_overflows, flush=True)
