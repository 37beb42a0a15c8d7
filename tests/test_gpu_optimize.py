"""hipexec_optimize: the hierarchy rebuilt with the surface-area topology made on the host (rt_bvh_host.hpp).

No counterpart in the reference (it has no hierarchy).  The tree only decides which exact tests are evaluated, so
everything must stay as it was, bit for bit: batch hits (id, t, u, v), any-hit answers, frames (ids and colours), ray
counts — against the executor's own answers before the call, against the index-order scans and against the oracle.
What must change is the work: fewer node visits per ray on the terrain scene."""
import numpy as np
import pytest

from raytracer_amd import _abi, scenegen
from tests.test_gpu_cull import _adversarial_rays, _nested_scene
from tests.util import ulp_diff

pytestmark = pytest.mark.gpu

CPU, CL = _abi.HIPEXEC_MODE_CPU, _abi.HIPEXEC_MODE_CL


def _answers(ex, rays, w, h, depth):
    out = []
    for mode in (CPU, CL):
        hit = ex.closest_hits(rays, mode)
        out += [hit["id"].copy()] + [hit[f].view(np.int32).copy() for f in ("t", "u", "v")] + [ex.any_hits(rays, mode) != 0]
    fr = ex.render(w, h, mode=CPU, max_depth=depth, tone=False)
    out += [fr["ids"].copy(), fr["rgb"].view(np.int32).copy()]
    return out, fr


@pytest.mark.parametrize("name", ["c1", "c2", "c4", "c5", "soup", "ties", "glass", "city", "city_yaw"])
def test_optimized_hierarchy_gives_the_same_bits_and_the_oracles(name):
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    fs = scenegen.small_variant(name).to_flat()
    rays = _adversarial_rays(fs, 77, n_each=300)
    orc = Oracle(fs)
    with HipExecutor(fs, device=0) as ex:
        before, fr0 = _answers(ex, rays, 96, 64, 3)
        ms = ex.optimize()
        assert ms > 0.0 or fs.n_tri <= 2
        assert ex.optimize() == 0.0                       # a second call has nothing to do
        after, fr1 = _answers(ex, rays, 96, 64, 3)
        for k, (a, b) in enumerate(zip(before, after)):
            assert np.array_equal(a, b), f"answer {k} changed with the rebuilt hierarchy"
        assert fr0["stats"].rays == fr1["stats"].rays
        assert fr1["stats"].bvh_stack_overflows == 0 or name == "soup"
        ex.set_algorithm("scan")
        scan, _ = _answers(ex, rays, 96, 64, 3)
        for k, (a, b) in enumerate(zip(scan, after)):
            assert np.array_equal(a, b), f"answer {k}: the rebuilt hierarchy and the scans differ"
    for mode, off in ((CPU, 0), (CL, 5)):
        ref = orc.closest_hits(rays, mode)
        assert np.array_equal(after[off], ref["id"])
        assert np.array_equal(after[off + 1], ref["t"].view(np.int32))
        assert np.array_equal(after[off + 4], orc.any_hits(rays, mode) != 0)
    ref = orc.render(96, 64, mode=CPU, max_depth=3, tone=0, n_threads=8)
    assert np.array_equal(fr1["ids"], ref["ids"])
    assert ulp_diff(fr1["rgb"], ref["linear"]).max() <= 4          # the tolerance of every frame test: linear colour within 4 ULP
    assert fr1["stats"].rays == ref["stats"].rays


def test_clones_see_the_rebuilt_hierarchy_and_visit_fewer_nodes():
    """The arrays are shared: executors cloned before and after the call all walk the new tree.  Terrain of 20 000
    triangles, depth 2: the frame's node visits drop, the frames stay the same."""
    from raytracer_amd.executor import HipExecutor
    fs = scenegen.scene_c2(seed=5, cells=100, tex_size=64).to_flat()
    with HipExecutor(fs, device=0) as ex:
        early = ex.clone()
        f0 = ex.render(320, 200, mode=CPU, max_depth=2, tone=False)
        ex.optimize()
        late = ex.clone()
        frames = [e.render(320, 200, mode=CPU, max_depth=2, tone=False) for e in (ex, early, late)]
        early.close(); late.close()
    for f in frames:
        assert np.array_equal(f["ids"], f0["ids"]) and np.array_equal(f["rgb"].view(np.int32), f0["rgb"].view(np.int32))
        assert sum(f["stats"].bvh_nodes) == sum(frames[0]["stats"].bvh_nodes)
    assert sum(frames[0]["stats"].bvh_nodes) < sum(f0["stats"].bvh_nodes)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 9])
def test_tiny_scenes_optimized(n):
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    fs = _nested_scene(n_levels=n - 1)
    rng = np.random.default_rng(n)
    o = rng.uniform([-3, -3, 10], [3, 3, 30], (500, 3))
    tgt = rng.uniform([-5, -5, -6], [5, 5, -2], (500, 3))
    rays = np.concatenate([o, tgt - o], 1).astype(np.float32)
    orc = Oracle(fs)
    with HipExecutor(fs, device=0) as ex:
        ex.optimize()
        ref = orc.closest_hits(rays, CPU); got = ex.closest_hits(rays, CPU)
        assert np.array_equal(got["id"], ref["id"])
        assert np.array_equal(ex.any_hits(rays, CPU) != 0, orc.any_hits(rays, CPU) != 0)


@pytest.mark.parametrize("seed,n_tri,extent", [(201, 300, 1.0e-2), (202, 5000, 10.0), (203, 40000, 50.0), (204, 20000, 3.0e4)])
def test_random_soups_optimized_equals_scan(seed, n_tri, extent):
    """Soups (threads of the host build are used from 16 384 triangles up): rebuilt hierarchy == index-order scans."""
    from raytracer_amd.executor import HipExecutor
    fs = scenegen.scene_soup(seed=seed, n_tri=n_tri, extent=extent).to_flat()
    rays = _adversarial_rays(fs, seed, n_each=250)
    with HipExecutor(fs, device=0) as ex:
        ex.optimize()
        res = {}
        for algo in ("bvh", "scan"):
            ex.set_algorithm(algo)
            per = []
            for mode in (CPU, CL):
                h = ex.closest_hits(rays, mode)
                per += [h["id"].copy(), h["t"].view(np.int32).copy(), (ex.any_hits(rays, mode) != 0)]
            fr = ex.render(96, 64, mode=CPU, max_depth=3, tone=False, algo=algo)
            per += [fr["ids"].copy(), fr["rgb"].view(np.int32).copy()]
            res[algo] = per
        for k, (a, b) in enumerate(zip(res["bvh"], res["scan"])):
            assert np.array_equal(a, b), f"result {k}: {np.count_nonzero(a != b)} entries differ"


@pytest.mark.parametrize("yaw", [False, True])
def test_orientation_classes_keep_mixed_scenes_culled(yaw):
    """Boxes on a ground grid: walls, roofs and ground in every neighbourhood.  A subtree that mixes them has no normal cone,
    its boxes take the loose enlargement and nothing is culled (without the classes: 1000 and more node visits per ray,
    millions of stack overflows), so both builds part the orientation classes under the root.  The frame must be the
    oracle's on a grid of blocks, identical for both builds and for the build without classes, with few node visits and no
    stack overflow.  `yaw`: the boxes stand at random angles, so the walls' normals fill the horizontal directions (normal
    cells of the large triangles instead of three axis classes; culled less well, see the bounds below)."""
    import os
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    fs = scenegen.scene_city(cells=48, n_boxes=400, yaw=yaw).to_flat()
    W, H, depth = 320, 200, 2
    with HipExecutor(fs, device=0) as ex:
        a = ex.render(W, H, mode=CPU, max_depth=depth, tone=False)
        ex.optimize()
        b = ex.render(W, H, mode=CPU, max_depth=depth, tone=False)
    os.environ["HIPEXEC_CLASS_BITS"] = "0"
    try:
        with HipExecutor(fs, device=0) as ex:
            c = ex.render(W, H, mode=CPU, max_depth=depth, tone=False)
    finally:
        os.environ.pop("HIPEXEC_CLASS_BITS", None)
    for f in (b, c):
        assert np.array_equal(f["ids"], a["ids"]) and np.array_equal(f["rgb"].view(np.int32), a["rgb"].view(np.int32))
    for f in (a, b):
        st = f["stats"]
        assert st.bvh_stack_overflows == 0
        assert sum(st.bvh_nodes) < (600 if yaw else 150) * st.rays, "the hierarchy does not cull"
    assert sum(c["stats"].bvh_nodes) > (1.5 if yaw else 3) * sum(a["stats"].bvh_nodes)
    orc = Oracle(fs)
    for (x, y) in ((0, 0), (152, 96), (312, 192), (96, 40), (224, 152)):
        ref = orc.render_region(W, H, x, y, 8, 8, mode=CPU, max_depth=depth, tone=0, n_threads=1)
        assert np.array_equal(a["ids"][y:y + 8, x:x + 8], ref["ids"]), f"hit ids differ in block {(x, y)}"
        assert ulp_diff(a["rgb"][y:y + 8, x:x + 8], ref["linear"]).max() <= 4
