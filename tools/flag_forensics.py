"""Any-hit and closest-hit answers of a given build of the library against the CPU oracle on the triangle-soup scenes:
python tools/flag_forensics.py <path to libhip_executor*.so>.  Used to compare the build whose per-lane traversal flags
are `bool` (-DRT_FLAG_BOOL) with the shipped one (32-bit words)."""
import os, sys
import os as _os; _os.environ.setdefault("HIPEXEC_KNOBS", "1")   # the HIPEXEC_* experiment knobs are honoured only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import _abi
lib = sys.argv[1]
_abi._lib = _abi.load_library(lib)
from raytracer_amd import scenegen
from raytracer_amd.executor import HipExecutor
from oracle.binding import Oracle
from tests.test_gpu_cull import _adversarial_rays
bad_total = 0
for (seed, n_tri, extent) in [(9, 1500, 10.0), (101, 12, 10.0), (104, 5000, 10.0), (107, 700, 0.3), (105, 40000, 50.0)]:
    fs = scenegen.scene_soup(seed=seed, n_tri=n_tri, extent=extent).to_flat()
    rays = _adversarial_rays(fs, seed, n_each=400)
    orc = Oracle(fs)
    with HipExecutor(fs, device=0) as ex:
        for mode in (0, 1):
            a = ex.any_hits(rays, mode) != 0
            r = orc.any_hits(rays, mode) != 0
            h = ex.closest_hits(rays, mode)["id"]
            hr = orc.closest_hits(rays, mode)["id"]
            fr = ex.render(96, 64, mode=0, max_depth=3, tone=False)
            rr = orc.render(96, 64, mode=0, max_depth=3, tone=0, n_threads=8)
            bad = int((a != r).sum()), int((h != hr).sum()), int((fr["ids"] != rr["ids"]).sum()), int(fr["stats"].rays_shadow != rr["stats"].rays_shadow)
            lost = int((r & ~a).sum())
            print(f"{os.path.basename(lib)} soup seed {seed} n_tri {n_tri} mode {mode}: any-hit mismatches {bad[0]} of {rays.shape[0]} (occluders lost: {lost}), "
                  f"closest-hit mismatches {bad[1]}, frame id mismatches {bad[2]}, linear colour max ulp "
                  f"{int(np.abs(fr['rgb'].view(np.int32).astype(np.int64) - rr['linear'].view(np.int32).astype(np.int64)).max())}", flush=True)
            bad_total += bad[0] + bad[1] + bad[2]
print("TOTAL MISMATCHES", bad_total)
