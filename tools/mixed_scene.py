"""Walk statistics and frame times on a scene whose neighbourhoods mix orientations (scenegen.scene_city: boxes on a ground
grid), for the device's hierarchy and the one hipexec_optimize makes: python tools/mixed_scene.py [n_boxes]"""
import os, sys, time
os.environ.setdefault("HIPEXEC_KNOBS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import scenegen
from raytracer_amd.executor import HipExecutor

n_boxes = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
yaw = len(sys.argv) > 2 and sys.argv[2] == "yaw"
fs = scenegen.scene_city(n_boxes=n_boxes, yaw=yaw).to_flat()
W, H, depth = 1280, 720, 2
ref = None
for how, bits, opt in (("device build, no orientation classes", "0", False), ("device build, three axis classes", "1", False),
                       ("device build, normal cells for large triangles", "2", False), ("hipexec_optimize, three axis classes", "1", True),
                       ("hipexec_optimize, normal cells for large triangles", "2", True)):
    os.environ["HIPEXEC_CLASS_BITS"] = bits
    with HipExecutor(fs, device=0) as ex:
        if opt:
            ms = ex.optimize()
        for _ in range(2):
            fr = ex.render(W, H, max_depth=depth, tone=False, time_kernels=False)
        t0 = time.perf_counter()
        for _ in range(5):
            fr = ex.render(W, H, max_depth=depth, tone=False, want_ids=True)
        dt = (time.perf_counter() - t0) / 5 * 1e3
        st = fr["stats"]
        rays = (st.rays_primary, st.rays_bounce, st.rays_shadow)
        walk = [(st.bvh_nodes[k] / max(1, rays[k]), st.bvh_tests[k] / max(1, rays[k])) for k in range(3)]
        print(f"{how}: {fs.n_tri} triangles, {W}x{H} depth {depth}: {dt:.2f} ms per frame incl. copy, device {st.total_ms:.2f} ms; nodes / exact tests per ray: "
              f"camera {walk[0][0]:.1f} / {walk[0][1]:.1f}, bounce {walk[1][0]:.1f} / {walk[1][1]:.1f}, shadow {walk[2][0]:.1f} / {walk[2][1]:.1f}; overflows {st.bvh_stack_overflows}", flush=True)
        if ref is None:
            ref = (fr["ids"].copy(), fr["rgb"].view(np.int32).copy())
        else:
            assert np.array_equal(ref[0], fr["ids"]) and np.array_equal(ref[1], fr["rgb"].view(np.int32)), "frames differ between the builds"
