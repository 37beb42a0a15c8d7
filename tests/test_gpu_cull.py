"""The hierarchy, the chunk bounding-box test and the edge-function filter only ever skip work;
none of them may change a result.  These tests aim rays at the places where a conservative bound is most likely
to be wrong (axis-parallel directions, origins on box faces and triangle planes, huge and tiny
direction lengths, far-away origins, unordered geometry with loose boxes) and compare with the
oracle bit for bit, and also compare the four on/off combinations of the two accelerations."""
import os
import numpy as np
import pytest

from raytracer_amd import _abi, scenegen

pytestmark = pytest.mark.gpu
CPU, CL = _abi.HIPEXEC_MODE_CPU, _abi.HIPEXEC_MODE_CL


def _adversarial_rays(fs, seed, n_each=400):
    rng = np.random.default_rng(seed)
    p = np.asarray(fs.tri_p, np.float32).reshape(-1, 3)
    e1 = np.asarray(fs.tri_e1, np.float32).reshape(-1, 3)
    e2 = np.asarray(fs.tri_e2, np.float32).reshape(-1, 3)
    nt = p.shape[0]
    lo, hi = p.min(0) - 1.0, p.max(0) + 1.0
    out = []
    # 1. axis-parallel rays (two direction components exactly zero), origins on a lattice that
    #    includes coordinates of triangle vertices (so rays run inside box faces)
    for ax in range(3):
        k = rng.integers(0, nt, n_each)
        o = p[k] + (rng.integers(0, 2, (n_each, 1)) * e1[k]) + (rng.integers(0, 2, (n_each, 1)) * e2[k])
        o[:, ax] = rng.choice([lo[ax], hi[ax]], n_each)
        d = np.zeros((n_each, 3), np.float32)
        d[:, ax] = np.where(o[:, ax] < 0.5 * (lo[ax] + hi[ax]), 1.0, -1.0) * rng.choice([1e-6, 1.0, 3e5], n_each)
        out.append(np.concatenate([o, d], 1))
    # 2. rays aimed exactly at vertices, edge midpoints and centroids from random origins
    k = rng.integers(0, nt, 3 * n_each)
    w = rng.choice([0.0, 0.5, 1.0 / 3.0, 1.0], (3 * n_each, 2)).astype(np.float32)
    tgt = p[k] + w[:, :1] * e1[k] + w[:, 1:] * e2[k]
    o = rng.uniform(lo, hi, (3 * n_each, 3)).astype(np.float32)
    out.append(np.concatenate([o, (tgt - o) * rng.choice([1e-3, 1.0, 50.0], (3 * n_each, 1))], 1))
    # 3. origins on a triangle's plane / inside the geometry, directions within the plane
    k = rng.integers(0, nt, n_each)
    o = p[k] + 0.3 * e1[k] + 0.3 * e2[k]
    d = rng.normal(size=(n_each, 1)) * e1[k] + rng.normal(size=(n_each, 1)) * e2[k]
    out.append(np.concatenate([o, d], 1))
    # 4. far-away origins looking back at the scene, and rays pointing away from everything
    c = 0.5 * (lo + hi)
    dirs = rng.normal(size=(n_each, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    o = c + dirs * rng.choice([1e2, 1e4, 1e6], (n_each, 1))
    tgt = rng.uniform(lo, hi, (n_each, 3))
    sign = rng.choice([1.0, -1.0], (n_each, 1), p=[0.8, 0.2])
    out.append(np.concatenate([o, sign * (tgt - o)], 1))
    # 5. nearly axis-parallel: one dominant component, others denormal-small or 1e-30
    o = rng.uniform(lo, hi, (n_each, 3))
    d = rng.choice([1e-30, -1e-30, 1e-42, 0.0], (n_each, 3))
    d[np.arange(n_each), rng.integers(0, 3, n_each)] = rng.choice([-1.0, 1.0], n_each)
    out.append(np.concatenate([o, d], 1))
    return np.ascontiguousarray(np.concatenate(out, 0), dtype=np.float32)


@pytest.fixture(scope="module")
def soup():
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    made = {}

    def get(name):
        if name not in made:
            fs = scenegen.small_variant(name).to_flat()
            made[name] = (fs, HipExecutor(fs, device=0), Oracle(fs))
        return made[name]
    yield get
    for _, ex, _ in made.values():
        ex.close()


@pytest.mark.parametrize("algo", ["bvh", "scan"])
@pytest.mark.parametrize("mode", [CPU, CL])
@pytest.mark.parametrize("name", ["soup", "c2", "c5", "ties", "city", "city_yaw"])
def test_adversarial_rays_match_oracle(soup, name, mode, algo):
    fs, ex, orc = soup(name)
    ex.set_algorithm(algo)
    rays = _adversarial_rays(fs, 21)
    ref = orc.closest_hits(rays, mode)
    got = ex.closest_hits(rays, mode)
    bad = np.nonzero(got["id"] != ref["id"])[0]
    assert bad.size == 0, f"{bad.size} closest hits differ, first ray {bad[:5]}: {rays[bad[:5]]}"
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f].view(np.int32), ref[f].view(np.int32)), f
    a_ref = orc.any_hits(rays, mode)
    a_got = ex.any_hits(rays, mode)
    bad = np.nonzero((a_got != 0) != (a_ref != 0))[0]
    assert bad.size == 0, f"{bad.size} any-hit answers differ, first ray {bad[:5]}: {rays[bad[:5]]}"


def test_soup_frame_matches_oracle(soup):
    fs, ex, orc = soup("soup")
    for mode, depth in ((CPU, 3), (CL, 0)):
        ref = orc.render(64, 48, mode=mode, max_depth=depth, tone=0, n_threads=8)
        for algo in ("bvh", "scan"):
            got = ex.render(64, 48, mode=mode, max_depth=depth, tone=False, algo=algo)
            assert np.array_equal(got["ids"], ref["ids"]), algo
            assert got["stats"].rays == ref["stats"].rays


def test_accelerations_do_not_change_results(soup):
    """Same scene, same rays: box culling on/off x edge filter on/off give identical bits."""
    from raytracer_amd.executor import HipExecutor
    results = []
    saved = {k: os.environ.get(k) for k in ("HIPEXEC_NO_CULL", "HIPEXEC_NO_FILTER")}
    try:
        for name in ("soup", "c2"):
            fs = scenegen.small_variant(name).to_flat()
            rays = _adversarial_rays(fs, 5, n_each=150)
            per = []
            for no_cull in ("0", "1"):
                for no_filter in ("0", "1"):
                    os.environ["HIPEXEC_NO_CULL"] = no_cull
                    os.environ["HIPEXEC_NO_FILTER"] = no_filter
                    with HipExecutor(fs, device=0) as ex:
                        ex.set_algorithm("scan")
                        h = ex.closest_hits(rays, CPU)
                        a = ex.any_hits(rays, CPU)
                        fr = ex.render(72, 40, mode=CPU, max_depth=2, tone=False, algo="scan")
                        per.append((h["id"].copy(), h["t"].view(np.int32).copy(), a.copy(),
                                    fr["ids"].copy(), fr["rgb"].view(np.int32).copy(),
                                    list(fr["stats"].scanned_chunks)))
            results.append(per)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for (name, per) in zip(("soup", "c2"), results):
        # the hierarchy gives the same bits as every scan configuration
        fs = scenegen.small_variant(name).to_flat()
        rays = _adversarial_rays(fs, 5, n_each=150)
        with HipExecutor(fs, device=0) as ex:
            h = ex.closest_hits(rays, CPU)
            a = ex.any_hits(rays, CPU)
            fr = ex.render(72, 40, mode=CPU, max_depth=2, tone=False, algo="bvh")
            assert fr["stats"].algorithm == _abi.HIPEXEC_ALGO_BVH and sum(fr["stats"].bvh_nodes) > 0
            per.append((h["id"].copy(), h["t"].view(np.int32).copy(), a.copy(), fr["ids"].copy(),
                        fr["rgb"].view(np.int32).copy(), None))
        base = per[0]
        for other in per[1:]:
            for x, y in zip(base[:5], other[:5]):
                assert np.array_equal(x, y)
        # culling really was active in the first configuration and off in the third
        assert sum(per[0][5]) < sum(per[2][5])


def _nested_scene(n_levels=30, L=4.0):
    """Large, mutually overlapping triangles whose centroids sit at 2^-k fractions of the scene cube:
    the Morton tree peels one triangle per level, so it is about n_levels deep, and a ray through
    the middle enters both children at every level -- deeper than the LDS traversal stack."""
    from raytracer_amd.scene import FlatScene, normalize_f32, cross_f32
    ext = 2 * L + 1.0
    cents = []
    for k in range(n_levels):
        c = np.zeros(3)
        c[k % 3] = ext * 2.0 ** -(k // 3 + 1)
        cents.append(c - L)
    cents.append(np.zeros(3) - L)
    cents = np.asarray(cents)
    n = cents.shape[0]
    rng = np.random.default_rng(4)
    tilt = rng.uniform(-0.3, 0.3, (n, 2))
    p = np.zeros((n, 3), np.float32); e1 = np.zeros((n, 3), np.float32); e2 = np.zeros((n, 3), np.float32)
    for i in range(n):
        a = np.array([-L, -L, tilt[i, 0]]); b = np.array([L, -L, tilt[i, 1]]); c = np.array([0.0, 2 * L, -tilt[i, 0] - tilt[i, 1]])
        shift = cents[i] - (a + b + c) / 3.0 + np.array([0, 0, 0.0])
        a, b, c = a + shift, b + shift, c + shift
        p[i] = a; e1[i] = b - a; e2[i] = c - a
    fs = FlatScene()
    fs.tri_p, fs.tri_e1, fs.tri_e2 = p, e1, e2
    fs.tri_norm = normalize_f32(cross_f32(e1, e2))
    fs.tri_uv = np.zeros((n, 6), np.float32); fs.tri_mat = np.zeros((n,), np.int32)
    fs.mat_color = [[0.7, 0.6, 0.5]]; fs.mat_kd = [0.8]; fs.mat_ks = [0.3]; fs.mat_hard = [20.0]; fs.mat_kr = [0.3]
    fs.mat_tex = np.asarray([-1], np.int32)
    fs.lamp_pos = [[2, 3, 20]]; fs.lamp_intensity = [1.0]; fs.lamp_distance = [30.0]
    fs.cam_pos = np.asarray([0.3, 0.2, 25.0], np.float32)
    return fs.normalised()


def test_stack_overflow_falls_back_to_the_same_answer():
    """Rays that outgrow the traversal stack are finished by a linear scan; results must not change.
    The stack is shortened to 3 entries through the executor's test hook so that most rays overflow."""
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    from tests.util import ulp_diff
    saved = os.environ.get("HIPEXEC_BVH_STACK_LIMIT")
    os.environ["HIPEXEC_BVH_STACK_LIMIT"] = "3"
    try:
        for fs in (_nested_scene(), scenegen.small_variant("c2").to_flat(), scenegen.small_variant("soup").to_flat()):
            rays = _adversarial_rays(fs, 13, n_each=120)
            orc = Oracle(fs)
            with HipExecutor(fs, device=0) as ex:
                for mode in (CPU, CL):
                    ref = orc.closest_hits(rays, mode)
                    got = ex.closest_hits(rays, mode)
                    assert np.array_equal(got["id"], ref["id"])
                    assert np.array_equal(got["t"].view(np.int32), ref["t"].view(np.int32))
                    assert np.array_equal(ex.any_hits(rays, mode) != 0, orc.any_hits(rays, mode) != 0)
                r = orc.render(48, 32, mode=CPU, max_depth=2, tone=0, n_threads=8)
                g = ex.render(48, 32, mode=CPU, max_depth=2, tone=False, algo="bvh")
                assert g["stats"].bvh_stack_overflows > 0
                assert np.array_equal(g["ids"], r["ids"])
                assert ulp_diff(g["rgb"], r["linear"]).max() <= 4
    finally:
        if saved is None:
            os.environ.pop("HIPEXEC_BVH_STACK_LIMIT", None)
        else:
            os.environ["HIPEXEC_BVH_STACK_LIMIT"] = saved


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 9])
def test_tiny_scenes(n):
    """Root-is-a-leaf and nearly empty hierarchies."""
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    fs = _nested_scene(n_levels=n - 1)
    rng = np.random.default_rng(n)
    o = rng.uniform([-3, -3, 10], [3, 3, 30], (500, 3))
    tgt = rng.uniform([-5, -5, -6], [5, 5, -2], (500, 3))
    rays = np.concatenate([o, tgt - o], 1).astype(np.float32)
    orc = Oracle(fs)
    with HipExecutor(fs, device=0) as ex:
        ref = orc.closest_hits(rays, CPU); got = ex.closest_hits(rays, CPU)
        assert np.array_equal(got["id"], ref["id"])
        assert np.array_equal(ex.any_hits(rays, CPU) != 0, orc.any_hits(rays, CPU) != 0)
        r = orc.render(40, 24, mode=CPU, max_depth=1, tone=0, n_threads=4)
        g = ex.render(40, 24, mode=CPU, max_depth=1, tone=False)
        assert np.array_equal(g["ids"], r["ids"])


@pytest.mark.parametrize("seed,n_tri,extent", [(101, 12, 10.0), (102, 300, 1.0e-2), (103, 300, 3.0e4), (104, 5000, 10.0),
                                              (105, 40000, 50.0), (106, 2000, 1.0e3), (107, 700, 0.3)])
def test_random_soups_bvh_equals_scan(seed, n_tri, extent):
    """Triangle soups of different sizes and scales (the enlargement of the boxes has absolute and
    relative terms): the hierarchy and the index-order scans must return the same bits for adversarial
    rays and for a frame; the small ones are also checked against the oracle."""
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    fs = scenegen.scene_soup(seed=seed, n_tri=n_tri, extent=extent).to_flat()
    rays = _adversarial_rays(fs, seed, n_each=250)
    with HipExecutor(fs, device=0) as ex:
        res = {}
        for algo in ("bvh", "scan"):
            ex.set_algorithm(algo)
            per = []
            for mode in (CPU, CL):
                h = ex.closest_hits(rays, mode)
                per += [h["id"].copy(), h["t"].view(np.int32).copy(), h["u"].view(np.int32).copy(), h["v"].view(np.int32).copy(),
                        (ex.any_hits(rays, mode) != 0)]
            fr = ex.render(96, 64, mode=CPU, max_depth=3, tone=False, algo=algo)
            per += [fr["ids"].copy(), fr["rgb"].view(np.int32).copy()]
            res[algo] = per                 # (the 40000-triangle soup overflows the traversal stack for ~100 rays: same bits)
        for k, (a, b) in enumerate(zip(res["bvh"], res["scan"])):
            assert np.array_equal(a, b), f"result {k}: {np.count_nonzero(a != b)} entries differ between the hierarchy and the scan"
        if n_tri <= 5000:
            orc = Oracle(fs)
            for mode, off in ((CPU, 0), (CL, 5)):
                ref = orc.closest_hits(rays, mode)
                assert np.array_equal(res["bvh"][off], ref["id"])
                assert np.array_equal(res["bvh"][off + 1], ref["t"].view(np.int32))
                assert np.array_equal(res["bvh"][off + 4], orc.any_hits(rays, mode) != 0)


def test_astronomic_coordinates_nan_candidates():
    """Triangles with coordinates around 1e17 (still below the bound up to which a hierarchy is built): the four
    determinants of the exact test overflow to inf and t = inf - inf = NaN passes every threshold comparison
    (`t < EPS` is false for NaN, ray_tracer.cpp:315).  The CPU tracer's scan then drops the candidate because it
    demands t > 0 (ray_tracer.cpp:214); the hierarchy must do the same, so that hierarchy, scan and oracle agree on
    every ray, including the ones whose only candidates are such NaN hits."""
    from raytracer_amd.executor import HipExecutor
    from raytracer_amd.scene import FlatScene
    from oracle.binding import Oracle
    rng = np.random.default_rng(77)
    n = 600
    scale = np.float32(1.0e17)
    c = rng.uniform(-1, 1, (n, 1, 3)) * scale
    v = (c + rng.normal(size=(n, 3, 3)) * scale * rng.choice([1e-3, 0.1, 1.0], (n, 1, 1))).astype(np.float32)
    fs = FlatScene()
    fs.tri_p = v[:, 0]; fs.tri_e1 = v[:, 1] - v[:, 0]; fs.tri_e2 = v[:, 2] - v[:, 0]
    with np.errstate(all="ignore"):
        from raytracer_amd.scene import cross_f32, normalize_f32
        fs.tri_norm = normalize_f32(cross_f32(fs.tri_e1, fs.tri_e2))
    fs.tri_mat = np.full(n, -1, np.int32)
    fs.cam_pos = np.zeros(3, np.float32)
    fs = fs.normalised()
    k = rng.integers(0, n, 4000)
    w = rng.uniform(0, 1, (4000, 2)).astype(np.float32)
    tgt = fs.tri_p[k] + w[:, :1] * 0.5 * fs.tri_e1[k] + w[:, 1:] * 0.5 * fs.tri_e2[k]
    o = (rng.uniform(-1.5, 1.5, (4000, 3)) * scale).astype(np.float32)
    d = ((tgt - o) * rng.choice([1e-17, 1e-3, 1.0], (4000, 1))).astype(np.float32)
    rays = np.ascontiguousarray(np.concatenate([o, d], 1), np.float32)
    orc = Oracle(fs)
    with HipExecutor(fs, device=0) as ex:
        for mode in (CPU, CL):
            ref = orc.closest_hits(rays, mode)
            res = {}
            for algo in ("bvh", "scan"):
                ex.set_algorithm(algo)
                res[algo] = ex.closest_hits(rays, mode)
                assert np.array_equal(res[algo]["id"], ref["id"]), (mode, algo, int((res[algo]["id"] != ref["id"]).sum()))
                assert np.array_equal(ex.any_hits(rays, mode) != 0, orc.any_hits(rays, mode) != 0), (mode, algo)
            assert np.array_equal(res["bvh"]["t"].view(np.int32), res["scan"]["t"].view(np.int32))
        assert (ref["id"] >= 0).any()


@pytest.mark.parametrize("log2_scale", [-44, -24, 20, 32, 40, 48])
def test_scaled_soups_axis_parallel_rays(log2_scale):
    """A soup scaled by 2^-44 ... 2^48 (exact in binary32, so the geometry is the same at every size).  The slab test of the
    hierarchy clamps the reciprocals of the direction to a bound that grows with the square of the scene's size and drops
    the clamp where that bound leaves the float range; axis-parallel and nearly axis-parallel rays (reciprocals at the
    clamp, or infinite without it) must get the oracle's answers in every regime."""
    from raytracer_amd.executor import HipExecutor
    from raytracer_amd.scene import FlatScene, cross_f32, normalize_f32
    from oracle.binding import Oracle
    base = scenegen.scene_soup(seed=131, n_tri=300, extent=10.0).to_flat()
    k = np.float32(2.0 ** log2_scale)
    fs = FlatScene()
    fs.tri_p = np.asarray(base.tri_p, np.float32).reshape(-1, 3) * k
    fs.tri_e1 = np.asarray(base.tri_e1, np.float32).reshape(-1, 3) * k
    fs.tri_e2 = np.asarray(base.tri_e2, np.float32).reshape(-1, 3) * k
    with np.errstate(all="ignore"):
        fs.tri_norm = normalize_f32(cross_f32(fs.tri_e1, fs.tri_e2))
    fs.tri_mat = np.full(fs.tri_p.shape[0], -1, np.int32)
    fs.cam_pos = np.zeros(3, np.float32)
    fs = fs.normalised()
    rays = _adversarial_rays(fs, 131 + log2_scale, n_each=200)
    # short directions too: |d| down to where the determinant threshold of the exact tests ends every hit
    rays = np.concatenate([rays, rays * np.array([1, 1, 1, 1e-9, 1e-9, 1e-9], np.float32)], 0)
    rays = np.ascontiguousarray(rays[np.isfinite(rays).all(1)], np.float32)
    orc = Oracle(fs)
    with HipExecutor(fs, device=0) as ex:
        hits = 0
        for mode in (CPU, CL):
            ref = orc.closest_hits(rays, mode); ref_any = orc.any_hits(rays, mode) != 0
            hits += int((ref["id"] >= 0).sum())
            for algo in ("bvh", "scan"):
                ex.set_algorithm(algo)
                got = ex.closest_hits(rays, mode)
                assert np.array_equal(got["id"], ref["id"]), (mode, algo, int((got["id"] != ref["id"]).sum()))
                assert np.array_equal(got["t"].view(np.int32), ref["t"].view(np.int32)), (mode, algo)
                assert np.array_equal(ex.any_hits(rays, mode) != 0, ref_any), (mode, algo)
        assert hits > 100 or log2_scale < 0        # (a tiny scene is below the determinant threshold for most directions)


@pytest.mark.parametrize("form", ["word", "bool"])
def test_lane_flag_forms_give_the_oracles_answers(form):
    """Regression test of the lane-flag episode: a traversal whose per-lane `found` flag was a `bool` lost occluders
    under -O3 (119 of 9563 shadow rays of the soup scene) while the same code with 0/1 words did not.  The shipped
    library uses words; its twin built with -DRT_FLAG_BOOL is the other form.  Both must give the oracle's any-hit and
    closest-hit answers and frames on the soups (shadow rays, bounce rays, stack-overflow fallback included).  If a
    kernel change ever exposes the defect again, the bool form fails here first."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "raytracer_amd", "lib", "libhip_executor.so" if form == "word" else "libhip_executor_boolflags.so")
    if not os.path.exists(lib):
        pytest.fail(f"{lib} is missing: run __graft_entry__.build()")
    # a process of its own: the library under test is chosen before the binding loads anything
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "flag_forensics.py"), lib], capture_output=True, text=True,
                       timeout=280, cwd=root)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "TOTAL MISMATCHES 0" in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("any-hit mismatches 0 of") == 10
