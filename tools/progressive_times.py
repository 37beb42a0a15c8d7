#!/usr/bin/env python3
"""Wall time of a C2 frame delivered to host memory: hipexec_render (whole frame, one copy) against
hipexec_render_progressive with several step sizes.  usage: python tools/progressive_times.py [workload]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                                   # noqa: E402
import bench                                                         # noqa: E402
from raytracer_amd import scenegen                                   # noqa: E402
from raytracer_amd.executor import HipExecutor                       # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
builder, kw, W, H, depth, desc = bench.WORKLOADS[wl]
fs = getattr(scenegen, builder)(**kw).to_flat()
img = np.zeros((H, W, 3), np.float32)
with HipExecutor(fs, device=0) as ex:
    def wall(fn, n=8):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n * 1e3
    full = wall(lambda: ex.render(W, H, max_depth=depth, out=img, want_ids=False))
    ref = img.copy()
    print(f"{wl}: hipexec_render {full:.2f} ms")
    for rows in (48, 96, 136, 272, 544):
        ms = wall(lambda: ex.render_progressive(W, H, rows_per_step=rows, max_depth=depth, out=img))
        same = bool(np.array_equal(img.view(np.int32), ref.view(np.int32)))
        print(f"{wl}: progressive, {rows:4d} rows per step ({-(-H // (-(-rows // 8) * 8))} steps): {ms:.2f} ms ({ms / full:.2f}x), same pixels: {same}")
