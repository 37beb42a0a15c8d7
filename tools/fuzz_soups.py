"""Randomised parity run: triangle soups of random size and scale (tests/test_gpu_cull.py's generator and adversarial rays,
other seeds), the hierarchy against the index-order scans bit for bit (closest hits, any hits, a depth-3 frame, both rule
sets) and both against the CPU oracle.  python tools/fuzz_soups.py [first_seed] [count]; stops at the first difference
and names its seed.  FUZZ_OPTIMIZE=1: the hierarchy is rebuilt with the surface-area topology first (hipexec_optimize)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import _abi, scenegen
from raytracer_amd.executor import HipExecutor
from oracle.binding import Oracle
from tests.test_gpu_cull import _adversarial_rays

CPU, CL = _abi.HIPEXEC_MODE_CPU, _abi.HIPEXEC_MODE_CL
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
n_rays = n_px = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n_tri = int(2 + 10.0 ** rng.uniform(0.3, 3.5))
    extent = float(10.0 ** rng.uniform(-3.0, 5.0))
    fs = scenegen.scene_soup(seed=seed, n_tri=n_tri, extent=extent).to_flat()
    rays = _adversarial_rays(fs, seed, n_each=120)
    W, H, depth = 56, 40, int(rng.integers(0, 4))
    orc = Oracle(fs)
    with HipExecutor(fs, device=0) as ex:
        if os.environ.get("FUZZ_OPTIMIZE") == "1":
            ex.optimize()
        res = {}
        for algo in ("bvh", "scan"):
            ex.set_algorithm(algo)
            per = []
            for mode in (CPU, CL):
                h = ex.closest_hits(rays, mode)
                per += [h["id"].copy(), h["t"].view(np.int32).copy(), h["u"].view(np.int32).copy(), h["v"].view(np.int32).copy(),
                        (ex.any_hits(rays, mode) != 0)]
            fr = ex.render(W, H, mode=CPU, max_depth=depth, tone=False, algo=algo)
            per += [fr["ids"].copy(), fr["rgb"].view(np.int32).copy(), int(fr["stats"].rays)]
            res[algo] = per
    for k, (a, b) in enumerate(zip(res["bvh"], res["scan"])):
        if not np.array_equal(a, b):
            raise SystemExit(f"seed {seed} (n_tri {n_tri}, extent {extent:g}): result {k} differs between the hierarchy and the scan")
    for mode, off in ((CPU, 0), (CL, 5)):
        ref = orc.closest_hits(rays, mode)
        ok = (np.array_equal(res["bvh"][off], ref["id"]) and np.array_equal(res["bvh"][off + 1], ref["t"].view(np.int32))
              and np.array_equal(res["bvh"][off + 4], orc.any_hits(rays, mode) != 0))
        if not ok:
            raise SystemExit(f"seed {seed} (n_tri {n_tri}, extent {extent:g}): batch results differ from the oracle, mode {mode}")
    ref = orc.render(W, H, mode=CPU, max_depth=depth, tone=0, n_threads=8)
    if not np.array_equal(res["bvh"][10], ref["ids"]) or int(ref["stats"].rays) != res["bvh"][12]:
        raise SystemExit(f"seed {seed} (n_tri {n_tri}, extent {extent:g}): frame ids or ray count differ from the oracle")
    d = np.abs(res["bvh"][11].astype(np.int64) - ref["linear"].view(np.int32).astype(np.int64)).max()
    if d > 4:
        raise SystemExit(f"seed {seed}: frame colour differs from the oracle by {d} ulp")
    n_rays += 2 * len(rays); n_px += W * H
    if (seed - first) % 10 == 9:
        print(f"seeds {first}..{seed} agree ({n_rays} batch rays, {n_px} pixels, {time.time() - t0:.0f} s)", flush=True)
print(f"{count} soups from seed {first}{' (surface-area topology)' if os.environ.get('FUZZ_OPTIMIZE') == '1' else ''}: hierarchy == scan bit for bit, both == oracle ({n_rays} batch rays x 2 rule sets x closest + any hit, "
      f"{n_px} pixels of depth 0-3 frames)")
