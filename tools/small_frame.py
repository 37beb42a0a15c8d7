"""Frame time of the reference's default workload (800x600, MODE_CL, one executor kept): python tools/small_frame.py [scene]"""
import os, sys, time
import os as _os; _os.environ.setdefault("HIPEXEC_KNOBS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracer_amd import scenegen, _abi
from raytracer_amd.executor import HipExecutor
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
fs = {"c2": scenegen.scene_c2, "c5": scenegen.scene_c5, "c4": scenegen.scene_c4, "c1": scenegen.scene_c1}[wl]().to_flat()
for (W, H, mode, depth, name) in ((800, 600, _abi.HIPEXEC_MODE_CL, 0, "800x600 CL"), (800, 600, _abi.HIPEXEC_MODE_CPU, 2, "800x600 CPU depth 2"),
                                   (640, 480, _abi.HIPEXEC_MODE_CPU, 0, "640x480 CPU depth 0")):
    out = torch.zeros((((H + 7) // 8) * 8, W, 4), dtype=torch.float32, device="cuda")
    with HipExecutor(fs, device=0) as ex:
        for _ in range(5):
            ex.render_device(W, H, out.data_ptr(), None, mode=mode, max_depth=depth); ex.sync()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); ex.render_device(W, H, out.data_ptr(), None, mode=mode, max_depth=depth); ex.sync(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"{wl} {name}: {ts[len(ts)//2]*1e3:.3f} ms per frame (median, device-resident, one frame at a time)", flush=True)
