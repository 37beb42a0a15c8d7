"""Time hipexec_create (scene upload + set-up kernels + hierarchy build) for a workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scenegen
from raytracer_amd.executor import HipExecutor
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
fs = {"c2": scenegen.scene_c2, "c5": scenegen.scene_c5, "c4": scenegen.scene_c4, "c1": scenegen.scene_c1}[wl]().to_flat()
for i in range(4):
    t0 = time.perf_counter()
    ex = HipExecutor(fs, device=0)
    t1 = time.perf_counter()
    ex.close()
    print(wl, "create ms", round((t1 - t0) * 1e3, 2), "destroy ms", round((time.perf_counter() - t1) * 1e3, 2), flush=True)
