"""Randomised parity run on the small variants of the benchmark scenes: random cameras (inside, around and far outside the
scene, any orientation), random depth, both rule sets, the optional modes where they apply, the hierarchy against the CPU
oracle (hit ids exact, linear colour within 4 ulp, ray counts equal) and, every fifth camera, the scans against the same
frame.  python tools/fuzz_cameras.py [first_seed] [cameras per scene]; stops at the first difference and names it.
FUZZ_OPTIMIZE=1: the hierarchy is rebuilt with the surface-area topology first (hipexec_optimize)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import _abi, scenegen
from raytracer_amd.executor import HipExecutor
from oracle.binding import Oracle

CPU, CL = _abi.HIPEXEC_MODE_CPU, _abi.HIPEXEC_MODE_CL
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
per_scene = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def rot(rng):
    a, b, c = rng.uniform(-np.pi, np.pi), rng.uniform(-1.2, 1.2), rng.uniform(-0.3, 0.3)
    ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    return (ry @ rx @ rz).astype(np.float32)


def ulp(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)).max()


t0 = time.time()
frames = 0
for name in ("c1", "c2", "c4", "c5", "mirror", "glass", "ties"):
    fs = scenegen.small_variant(name).to_flat()
    p = np.asarray(fs.tri_p, np.float32).reshape(-1, 3)
    lo, hi = p.min(0), p.max(0)
    with HipExecutor(fs, device=0) as ex:
        if os.environ.get("FUZZ_OPTIMIZE") == "1":
            ex.optimize()
        for k in range(per_scene):
            rng = np.random.default_rng(first * 1000 + k)
            where = rng.integers(0, 3)
            span = (hi - lo) * (0.5, 1.5, 40.0)[where]
            pos = (0.5 * (lo + hi) + rng.uniform(-1, 1, 3) * span + np.array([0, 0.5 * (hi[1] - lo[1]), 0])).astype(np.float32)
            mat = (fs.cam_mat @ rot(rng)).astype(np.float32)
            g = copy.copy(fs); g.cam_pos = pos.copy(); g.cam_mat = mat.copy()
            W, H = int(rng.integers(3, 9)) * 8, int(rng.integers(2, 7)) * 8 - int(rng.integers(0, 8))
            depth = int(rng.integers(0, 6))
            mode = CPU if rng.integers(0, 4) else CL
            opts = {}
            if mode == CPU and name == "glass" and rng.integers(0, 2):
                opts["refract"] = True
            if mode == CPU and name == "c4" and rng.integers(0, 2):
                opts["bilinear"] = True
            ex.set_camera(pos, mat)
            ref = Oracle(g).render(W, H, mode=mode, max_depth=depth, tone=0, n_threads=8, **opts)
            for algo in (("bvh", "scan") if k % 5 == 0 else ("bvh",)):
                got = ex.render(W, H, mode=mode, max_depth=depth, tone=False, algo=algo, **opts)
                what = f"{name} camera seed {first * 1000 + k} ({W}x{H}, depth {depth}, mode {mode}, {opts}, {algo})"
                if not np.array_equal(got["ids"], ref["ids"]):
                    raise SystemExit(what + ": hit ids differ from the oracle")
                if ulp(got["rgb"], ref["linear"]) > 4:
                    raise SystemExit(what + f": colour differs by {ulp(got['rgb'], ref['linear'])} ulp")
                for f in ("rays_primary", "rays_bounce", "rays_shadow"):
                    if getattr(got["stats"], f) != getattr(ref["stats"], f):
                        raise SystemExit(what + f": {f} {getattr(got['stats'], f)} against {getattr(ref['stats'], f)}")
                frames += 1
            if k % 100 == 99 and k + 1 < per_scene:
                print(f"{name}: {k + 1} cameras so far ({time.time() - t0:.0f} s)", flush=True)
    print(f"{name}: {per_scene} cameras agree ({frames} frames, {time.time() - t0:.0f} s)", flush=True)
print(f"{frames} frames of 7 scenes from seed {first}{' (surface-area topology)' if os.environ.get('FUZZ_OPTIMIZE') == '1' else ''}: ids exact, colour within 4 ulp, ray counts equal")
