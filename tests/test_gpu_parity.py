"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded scenes.  Bars: hit ids bit-exact; ray counts and algorithmic test counts equal;
linear colour within 4 ULP (the only non-bit-exact operation is powf, evaluated by different
libm implementations on host and device); RGB8 within +-1 LSB on >= 99.9 % of pixels and
PSNR >= 50 dB."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import _abi, scenegen
from tests.util import assert_image_parity, ulp_diff

pytestmark = pytest.mark.gpu

CPU, CL = _abi.HIPEXEC_MODE_CPU, _abi.HIPEXEC_MODE_CL


@pytest.fixture(scope="module")
def backends():
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    cache = {}

    def get(name):
        if name not in cache:
            fs = scenegen.small_variant(name).to_flat()
            cache[name] = (fs, HipExecutor(fs, device=0), Oracle(fs))
        return cache[name]
    yield get
    for _, ex, _ in cache.values():
        ex.close()


def _check_frame(ex, orc, W, H, mode, depth):
    """Both algorithms of the executor against the oracle."""
    ref = orc.render(W, H, mode=mode, max_depth=depth, tone=1, n_threads=8)
    rs = ref["stats"]
    for algo in ("bvh", "scan"):
        got = ex.render(W, H, mode=mode, max_depth=depth, tone=True, algo=algo)
        lin = ex.render(W, H, mode=mode, max_depth=depth, tone=False, want_ids=False, algo=algo)
        assert np.array_equal(got["ids"], ref["ids"]), f"{algo}: {(got['ids'] != ref['ids']).sum()} hit ids differ"
        gs = got["stats"]
        assert gs.algorithm == {"scan": _abi.HIPEXEC_ALGO_SCAN, "bvh": _abi.HIPEXEC_ALGO_BVH}[algo]
        keys = ["rays_primary", "rays_bounce", "rays_shadow", "tests_tri_closest", "tests_sph_closest", "tests_tri_primary"]
        if algo == "scan":                       # only the index-order scans know the reference's any-hit test counts
            keys += ["tests_tri_any", "tests_sph_any"]
        for k in keys:
            assert getattr(gs, k) == getattr(rs, k), (algo, k)
        u = ulp_diff(lin["rgb"], ref["linear"])
        assert u.max() <= 4, f"{algo}: linear colour off by {u.max()} ULP"
        assert_image_parity(got["rgb"], ref["rgb"])


@pytest.mark.parametrize("name", scenegen.SMALL_SCENES)
@pytest.mark.parametrize("depth", [0, 3])
def test_frame_cpu_mode(backends, name, depth):
    fs, ex, orc = backends(name)
    _check_frame(ex, orc, 64, 48, CPU, depth)


@pytest.mark.parametrize("name", scenegen.SMALL_SCENES)
def test_frame_cl_mode(backends, name):
    fs, ex, orc = backends(name)
    _check_frame(ex, orc, 64, 48, CL, 0)


def test_ragged_frame_and_depth5(backends):
    fs, ex, orc = backends("mirror")
    _check_frame(ex, orc, 61, 43, CPU, 5)      # width and height not multiples of 8
    _check_frame(ex, orc, 7, 5, CPU, 1)        # smaller than one tile
    _check_frame(ex, orc, 61, 43, CL, 0)


def test_tile_shards_compose(backends):
    """Interleaved tile shards (the multi-GPU partition) reproduce the unsharded frame bit for bit."""
    fs, ex, orc = backends("c2")
    W, H = 64, 52
    full = ex.render(W, H, max_depth=2)
    img = np.zeros((H, W, 3), np.float32)
    for G in (2, 3):
        img[:] = -1
        rays = 0
        for r in range(G):
            part = ex.render(W, H, max_depth=2, tile_first=r, tile_step=G, out=img, want_ids=False)
            rays += part["stats"].rays
        assert np.array_equal(img.view(np.int32), full["rgb"].view(np.int32))
        assert rays == full["stats"].rays


def _random_rays(fs, n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform([-8, 0.5, -14], [8, 8, 6], size=(n, 3))
    tgt = rng.uniform([-10, -0.5, -18], [10, 3, 2], size=(n, 3))
    d = tgt - o
    d *= rng.uniform(0.2, 3.0, size=(n, 1))          # un-normalised on purpose
    return np.concatenate([o, d], axis=1).astype(np.float32)


@pytest.mark.parametrize("algo", ["bvh", "scan"])
@pytest.mark.parametrize("mode", [CPU, CL])
@pytest.mark.parametrize("name", ["c2", "ties", "c5"])
def test_batch_entry_points(backends, name, mode, algo):
    fs, ex, orc = backends(name)
    ex.set_algorithm(algo)
    rays = _random_rays(fs, 3000, 11)
    h_ref = orc.closest_hits(rays, mode)
    h_got = ex.closest_hits(rays, mode)
    assert np.array_equal(h_got["id"], h_ref["id"])
    assert np.array_equal(h_got["mat"], h_ref["mat"])
    for f in ("t", "u", "v", "point", "norm", "color"):
        assert np.array_equal(h_got[f].view(np.int32), h_ref[f].view(np.int32)), f
    pre = (np.arange(rays.shape[0]) % 7 == 0).astype(np.int8)
    a_ref = orc.any_hits(rays, mode, hits=pre.copy())
    a_got = ex.any_hits(rays, mode, hits=pre.copy())
    assert np.array_equal(a_got != 0, a_ref != 0)
    base = np.full((rays.shape[0], 3), 0.25, np.float32)
    c_ref = orc.trace_rays(rays, mode, max_depth=2, colours=base.copy())
    c_got = ex.trace_rays(rays, mode, max_depth=2, colours=base.copy())
    ex.set_algorithm("bvh")
    assert ulp_diff(c_got, c_ref).max() <= 4


def test_empty_inputs(backends):
    from raytracer_amd.executor import HipExecutor
    from raytracer_amd.scene import FlatScene
    fs, ex, orc = backends("c1")
    z = np.zeros((0, 6), np.float32)
    assert ex.closest_hits(z).shape == (0,)
    assert ex.any_hits(z).shape == (0,)
    assert ex.trace_rays(z).shape == (0, 3)
    empty = FlatScene().normalised()                 # no triangles, spheres, lamps: everything misses
    with HipExecutor(empty, device=0) as e2:
        r = e2.render(16, 8, mode=CL)
        assert (r["ids"] == -1).all()
        assert np.allclose(r["rgb"], np.float32(np.power(np.float32(0.3) / np.float32(2.2), 0.3)), rtol=1e-6)
        rays = _random_rays(empty, 10, 3)
        assert (e2.closest_hits(rays)["id"] == -1).all()
        assert (e2.any_hits(rays) == 0).all()


def test_two_lane_frame(backends):
    """Tall frames with reflections are split over two concurrent lanes (even / odd tiles); the
    result and the counters are those of the oracle, including a ragged last tile."""
    fs, ex, orc = backends("mirror")
    W, H = 40, 260                                   # 33 tiles, the last one 4 rows high
    ref = orc.render(W, H, mode=CPU, max_depth=3, tone=0, n_threads=8)
    got = ex.render(W, H, mode=CPU, max_depth=3, tone=False, algo="bvh")
    assert got["stats"].lanes == 2
    assert np.array_equal(got["ids"], ref["ids"])
    assert ulp_diff(got["rgb"], ref["linear"]).max() <= 4
    for k in ("rays_primary", "rays_bounce", "rays_shadow", "tests_tri_closest"):
        assert getattr(got["stats"], k) == getattr(ref["stats"], k), k
    one = ex.render(W, H, mode=CPU, max_depth=0, tone=False, algo="bvh")       # no bounces: one lane
    assert one["stats"].lanes == 1
    # shards: 16 tiles still run on two lanes, 4 tiles are below the threshold; either way the shard's rows are the frame's
    for first, step, lanes in ((1, 2, 2), (3, 8, 1)):
        img = np.full((H, W, 3), -7.0, np.float32)
        part = ex.render(W, H, mode=CPU, max_depth=3, tone=False, algo="bvh", tile_first=first, tile_step=step, want_ids=False, out=img)
        assert part["stats"].lanes == lanes and part["stats"].rays < got["stats"].rays
        for t in range((H + 7) // 8):
            rows = slice(8 * t, min(8 * t + 8, H))
            if t >= first and (t - first) % step == 0:
                assert np.array_equal(img[rows].view(np.int32), got["rgb"][rows].view(np.int32)), t
            else:
                assert (img[rows] == -7.0).all(), t


@pytest.mark.parametrize("packet", ["0", "1"])
def test_camera_rays_as_packets_or_per_lane(packet, backends, monkeypatch):
    """HIPEXEC_PACKET=1 (default) walks the hierarchy once per 8x8-pixel tile of camera rays, 0 once per ray;
    hit ids and colours are the same (the walk order, and with it the number of tests the shrinking limit
    prunes, differs)."""
    from raytracer_amd.executor import HipExecutor
    monkeypatch.setenv("HIPEXEC_PACKET", packet)
    for name in ("c2", "soup", "mirror"):
        fs, ex_default, orc = backends(name)
        with HipExecutor(fs, device=0) as ex:
            for mode, depth in ((CPU, 2), (CL, 0)):
                ref = orc.render(61, 43, mode=mode, max_depth=depth, tone=0, n_threads=8)
                got = ex.render(61, 43, mode=mode, max_depth=depth, tone=False, algo="bvh")
                assert np.array_equal(got["ids"], ref["ids"]), name
                assert ulp_diff(got["rgb"], ref["linear"]).max() <= 4
                base = ex_default.render(61, 43, mode=mode, max_depth=depth, tone=False, algo="bvh")
                assert np.array_equal(got["rgb"].view(np.int32), base["rgb"].view(np.int32))


def test_camera_rays_made_by_the_packet_kernel_or_by_a_launch_of_their_own(backends, monkeypatch):
    """HIPEXEC_FUSE_RAYGEN=0 stores the camera rays with k_raygen before the packet kernel walks them; by default the
    packet kernel makes them itself.  Same expressions, hence the same frame bit for bit (ragged sizes included)."""
    from raytracer_amd.executor import HipExecutor
    for name in ("c2", "mirror"):
        fs = scenegen.small_variant(name).to_flat()
        monkeypatch.setenv("HIPEXEC_FUSE_RAYGEN", "0")
        ex = HipExecutor(fs, device=0)
        monkeypatch.setenv("HIPEXEC_FUSE_RAYGEN", "1")
        ex_default = HipExecutor(fs, device=0)
        with ex, ex_default:
            for mode, depth, W, H in ((CPU, 2, 61, 43), (CL, 0, 50, 33)):
                got = ex.render(W, H, mode=mode, max_depth=depth, tone=False, algo="bvh")
                base = ex_default.render(W, H, mode=mode, max_depth=depth, tone=False, algo="bvh")
                assert np.array_equal(got["ids"], base["ids"])
                assert np.array_equal(got["rgb"].view(np.int32), base["rgb"].view(np.int32))
                assert got["stats"].rays == base["stats"].rays
                assert got["stats"].launches[0] == base["stats"].launches[0] + 1      # k_raygen


def test_bounce_rays_one_per_lane_or_refilled(backends, monkeypatch):
    """By default a lane of the closest-hit kernel of the bounce rays traces one ray; with HIPEXEC_SIMPLE_FROM above the
    launch size the lanes of a block take the block's rays one at a time instead (k_bvh_closest_p).  Same hits, same frame."""
    from raytracer_amd.executor import HipExecutor
    monkeypatch.setenv("HIPEXEC_SIMPLE_FROM", "4000000000")
    for name in ("mirror", "soup"):
        fs, ex_default, orc = backends(name)
        with HipExecutor(fs, device=0) as ex:
            ref = orc.render(61, 83, mode=CPU, max_depth=3, tone=0, n_threads=8)
            got = ex.render(61, 83, mode=CPU, max_depth=3, tone=False, algo="bvh")
            base = ex_default.render(61, 83, mode=CPU, max_depth=3, tone=False, algo="bvh")
            assert np.array_equal(got["ids"], ref["ids"])
            assert np.array_equal(got["rgb"].view(np.int32), base["rgb"].view(np.int32))
            assert got["stats"].rays == ref["stats"].rays == base["stats"].rays


@pytest.mark.parametrize("mask", ["0", "1", "2"])
def test_uniform_steps_through_the_scalar_cache_or_per_lane(mask, backends, monkeypatch):
    """HIPEXEC_UNIFORM_FETCH: bit 0 lets the shadow rays, bit 1 the bounce and batch closest-hit rays fetch a node record
    through the scalar cache in the steps in which every lane of the wave sits on the same node (default: both).  Every
    combination gives the default's frame and hits bit for bit, and the oracle's."""
    from raytracer_amd.executor import HipExecutor
    monkeypatch.setenv("HIPEXEC_UNIFORM_FETCH", mask)
    for name in ("mirror", "soup", "c2"):
        fs, ex_default, orc = backends(name)
        with HipExecutor(fs, device=0) as ex:
            ref = orc.render(61, 83, mode=CPU, max_depth=3, tone=0, n_threads=8)
            got = ex.render(61, 83, mode=CPU, max_depth=3, tone=False, algo="bvh")
            base = ex_default.render(61, 83, mode=CPU, max_depth=3, tone=False, algo="bvh")
            assert np.array_equal(got["ids"], ref["ids"]), name
            assert np.array_equal(got["rgb"].view(np.int32), base["rgb"].view(np.int32)), name
            assert got["stats"].rays == ref["stats"].rays == base["stats"].rays
            rays = _random_rays(fs, 3000, 5)
            rays[:1500, :3] = rays[0, :3]                    # a bundle from one origin: uniform steps at the top of the tree
            for mode in (CPU, CL):
                a, b = ex.closest_hits(rays, mode), ex_default.closest_hits(rays, mode)
                for k in ("id", "t", "u", "v"):
                    assert np.array_equal(a[k].view(np.int32), b[k].view(np.int32)), (name, k)
                assert np.array_equal(a["id"], orc.closest_hits(rays, mode)["id"]), name


def test_more_spheres_than_the_shadow_kernel_keeps_in_lds():
    """The shadow-ray kernel keeps the first 64 spheres in LDS and reads the others from memory: 90 spheres, 3 lamps."""
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    fs = scenegen.scene_c5(cells=12, n_lamps=3, n_spheres=90).to_flat()
    orc = Oracle(fs)
    with HipExecutor(fs, device=0) as ex:
        for mode in (CPU, CL):
            ref = orc.render(72, 40, mode=mode, max_depth=0, tone=0, n_threads=8)
            got = ex.render(72, 40, mode=mode, max_depth=0, tone=False, algo="bvh")
            assert np.array_equal(got["ids"], ref["ids"])
            assert ulp_diff(got["rgb"], ref["linear"]).max() <= 4
            assert got["stats"].rays_shadow == ref["stats"].rays_shadow


def test_timing_by_kernel_class(backends):
    """HIPEXEC_FLAG_TIME_KERNELS times every launch; with HIPEXEC_FLAG_TIME_CLASS bits only the named classes."""
    fs, ex, orc = backends("c2")
    every = ex.render(64, 48, mode=CPU, max_depth=2, want_ids=False, time_kernels=True, algo="bvh")["stats"]
    assert all(every.kernel_ms[k] > 0.0 for k in range(4))
    only = ex.render(64, 48, mode=CPU, max_depth=2, want_ids=False, time_kernels=[_abi.HIPEXEC_K_ANYHIT], algo="bvh")["stats"]
    assert only.kernel_ms[_abi.HIPEXEC_K_ANYHIT] > 0.0
    assert all(only.kernel_ms[k] == 0.0 for k in range(4) if k != _abi.HIPEXEC_K_ANYHIT)
    assert list(only.launches) == list(every.launches) and only.rays == every.rays
    none = ex.render(64, 48, mode=CPU, max_depth=2, want_ids=False, algo="bvh")["stats"]
    assert all(none.kernel_ms[k] == 0.0 for k in range(4))


@pytest.mark.parametrize("direct", ["0", "1"])
@pytest.mark.parametrize("fuse", ["0", "2"])
@pytest.mark.parametrize("tail_max", ["0", "300000"])
@pytest.mark.parametrize("overlap", ["0", "2"])
def test_levels_on_one_or_two_streams(overlap, tail_max, fuse, direct, monkeypatch):
    """HIPEXEC_OVERLAP=2 lights every level on a side stream while the next one is traced, 0 keeps one
    stream per lane; HIPEXEC_TAIL_MAX=0 traces every level with its own launches, otherwise small ray
    queues are followed to the end of their paths by one kernel; HIPEXEC_FUSE_SHADE=2 shades inside the
    shadow-ray kernel, 0 in a launch of its own; HIPEXEC_SHADOW_DIRECT=1 lets every lane go through the lamps with
    its own sample, 0 compacts the block's (sample, lamp) pairs into a list first.  The frame, the hit ids and the ray counts are those of the
    oracle in every combination."""
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    monkeypatch.setenv("HIPEXEC_OVERLAP", overlap)
    monkeypatch.setenv("HIPEXEC_TAIL_MAX", tail_max)
    monkeypatch.setenv("HIPEXEC_FUSE_SHADE", fuse)
    monkeypatch.setenv("HIPEXEC_SHADOW_DIRECT", direct)
    fs = scenegen.small_variant("mirror").to_flat()
    W, H = 48, 132
    orc = Oracle(fs)
    with HipExecutor(fs, device=0) as ex:
        ref = orc.render(W, H, mode=CPU, max_depth=4, tone=0, n_threads=8)
        for _ in range(2):                                  # the second frame reuses every buffer and event
            got = ex.render(W, H, mode=CPU, max_depth=4, tone=False, algo="bvh")
            assert np.array_equal(got["ids"], ref["ids"])
            assert ulp_diff(got["rgb"], ref["linear"]).max() <= 4
            for k in ("rays_primary", "rays_bounce", "rays_shadow", "tests_tri_closest", "tests_sph_closest"):
                assert getattr(got["stats"], k) == getattr(ref["stats"], k), k
        rays = _random_rays(fs, 500, 11)
        assert ulp_diff(ex.trace_rays(rays, CPU, max_depth=3), orc.trace_rays(rays, CPU, max_depth=3)).max() <= 4


@pytest.mark.parametrize("mode,depth", [(CPU, 2), (CL, 0)])
def test_progressive_render_delivers_the_same_frame_in_steps(backends, mode, depth):
    """Progressive delivery (the role of renderParallel + the display loop, ray_tracer.cpp:57-153, main.cpp:53-90):
    after every step the delivered rows equal the ORACLE's rows (parity bar of the frame tests), later rows are
    still untouched; the finished image is also what hipexec_render gives, bit for bit."""
    fs, ex, orc = backends("c2")
    W, H = 56, 100                                   # 13 tiles, the last one 4 rows high
    ref = orc.render(W, H, mode=mode, max_depth=depth, tone=1, n_threads=8)
    img = np.full((H, W, 3), -7.0, np.float32)
    seen = []

    def on_rows(y0, rows):
        # the rows of this step are final, later rows are still untouched
        assert_image_parity(img[y0:y0 + rows], ref["rgb"][y0:y0 + rows], min_frac=1.0)
        assert (img[y0 + rows:] == -7.0).all()
        seen.append((y0, rows))

    got = ex.render_progressive(W, H, rows_per_step=20, on_rows=on_rows, mode=mode, max_depth=depth, out=img)
    assert seen == [(0, 24), (24, 24), (48, 24), (72, 24), (96, 4)]
    assert_image_parity(img, ref["rgb"], min_frac=1.0)
    for k in ("rays_primary", "rays_bounce", "rays_shadow"):
        assert getattr(got["stats"], k) == getattr(ref["stats"], k), k
    lin = np.zeros((H, W, 3), np.float32)
    ex._check(ex._lib.hipexec_render_progressive(ex._h, C.byref(ex._cfg(W, H, mode, depth, False, 0, 1, False)), 20,
                                                 lin.ctypes.data, None, None, None))
    assert ulp_diff(lin, ref["linear"]).max() <= 4
    full = ex.render(W, H, mode=mode, max_depth=depth, tone=True, want_ids=False)
    assert np.array_equal(img.view(np.int32), full["rgb"].view(np.int32))
    with pytest.raises(RuntimeError):
        ex._check(ex._lib.hipexec_render_progressive(ex._h, None, 8, img.ctypes.data, None, None, None))


@pytest.mark.parametrize("name", ["c2", "c4", "ties"])
def test_bilinear_mode_matches_its_cpu_definition(backends, name):
    """The optional bilinear texture mode (no reference counterpart) against the oracle's definition."""
    fs, ex, orc = backends(name)
    W, H = 64, 48
    for mode, depth in ((CPU, 2), (CL, 0)):
        ref = orc.render(W, H, mode=mode, max_depth=depth, tone=0, n_threads=8, bilinear=True)
        near = orc.render(W, H, mode=mode, max_depth=depth, tone=0, n_threads=8)
        for algo in ("bvh", "scan"):
            got = ex.render(W, H, mode=mode, max_depth=depth, tone=False, bilinear=True, algo=algo)
            assert np.array_equal(got["ids"], ref["ids"])
            assert ulp_diff(got["rgb"], ref["linear"]).max() <= 4
        assert not np.array_equal(ref["linear"], near["linear"])          # the mode does something
        back = ex.render(W, H, mode=mode, max_depth=depth, tone=False)     # and is off again afterwards
        assert ulp_diff(back["rgb"], near["linear"]).max() <= 4


@pytest.mark.parametrize("mode,depth", [(CL, 0), (CPU, 2)])
def test_sub_rectangles_equal_the_frame_and_the_oracle(backends, mode, depth):
    """hipexec_render_rect (the role of renderSubBlockCl, ray_tracer_cl.cpp:59-105): rectangles with unaligned
    origins and sizes, partly or wholly outside the frame; inside the rectangle the oracle's pixels and ids, outside
    it nothing is written; the 48x48 block walk of ray_tracer_cl.cpp:15-55 reassembles hipexec_render's frame."""
    fs, ex, orc = backends("c2")
    W, H = 107, 61
    spp = 1 if mode == CL else 4
    ref = orc.render(W, H, mode=mode, max_depth=depth, tone=1, n_threads=8)
    full = ex.render(W, H, mode=mode, max_depth=depth, tone=True)
    for (x, y, w, h) in [(0, 0, 48, 48), (48, 0, 48, 48), (96, 48, 11, 13), (5, 7, 1, 1), (13, 3, 70, 9), (100, 50, 48, 48),
                         (-5, -3, 20, 10), (0, 60, 107, 1), (200, 10, 5, 5), (3, 3, 0, 4)]:
        img = np.full((H, W, 3), -7.0, np.float32)
        ids = np.full((H, W, spp), -9, np.int32)
        st = ex.render_rect(W, H, x, y, w, h, img, mode=mode, max_depth=depth, ids=ids)
        x0, y0, x1, y1 = max(0, x), max(0, y), max(0, min(W, x + w)), max(0, min(H, y + h))
        inside = np.zeros((H, W), bool)
        if x1 > x0 and y1 > y0:
            inside[y0:y1, x0:x1] = True
        assert (img[~inside] == -7.0).all() and (ids[~inside] == -9).all()
        assert np.array_equal(img[inside].view(np.int32), full["rgb"][inside].view(np.int32))
        assert np.array_equal(ids[inside], ref["ids"][inside])
        assert st.rays_primary == int(inside.sum()) * spp
        if inside.any():
            assert_image_parity(img[inside], ref["rgb"][inside], min_frac=1.0)
    walk = np.zeros((H, W, 3), np.float32)
    rays = 0
    for by in range(0, H, 48):
        for bx in range(0, W, 48):
            rays += ex.render_rect(W, H, bx, by, min(48, W - bx), min(48, H - by), walk, mode=mode, max_depth=depth).rays
    assert np.array_equal(walk.view(np.int32), full["rgb"].view(np.int32))
    assert rays == full["stats"].rays == ref["stats"].rays
    with pytest.raises(RuntimeError):
        ex._check(ex._lib.hipexec_render_rect(ex._h, C.byref(ex._cfg(W, H, mode, depth, True, 1, 2, False)), 0, 0, 8, 8,
                                              walk.ctypes.data, None, None))


@pytest.mark.parametrize("spoil", ["0", "64", "-9000", "off", "bound"])
def test_remembered_queue_sizes(monkeypatch, spoil):
    """The launches of the recursion levels are sized on the host: from the bound n(L+1) <= n(L) in the first frame of a
    configuration, from the queue sizes the last frame produced afterwards, never waiting for a count; the produced
    counts are checked when the call is collected.  HIPEXEC_PREDICT_SPOIL stores wrong sizes: too large is harmless,
    too small by more than the margin (-9000 of level 1's 25 344 rays) loses rays on the device, which the check must
    catch (the frame is then traced again from the bound, and the call's statistics say so);
    the frame must come out right all the same.  HIPEXEC_PREDICT=0 always waits for the counts, =2 always sizes from
    the bound.  Synchronous, sharded and asynchronous (device-resident) frames against the oracle, three frames each."""
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    from tests.util import HipMem
    if spoil in ("off", "bound"):
        monkeypatch.setenv("HIPEXEC_PREDICT", "0" if spoil == "off" else "2")
    else:
        monkeypatch.setenv("HIPEXEC_PREDICT_SPOIL", spoil)
    fs = scenegen.small_variant("mirror").to_flat()
    W, H, depth = 48, 132, 4
    ref = Oracle(fs).render(W, H, mode=CPU, max_depth=depth, tone=0, n_threads=8)
    hip = HipMem()
    d_out = hip.malloc(((H + 7) // 8) * 8 * W * 16)
    try:
        with HipExecutor(fs, device=0) as ex:
            retraced = []
            for _ in range(3):
                got = ex.render(W, H, mode=CPU, max_depth=depth, tone=False)
                assert np.array_equal(got["ids"], ref["ids"])
                assert ulp_diff(got["rgb"], ref["linear"]).max() <= 4
                for k in ("rays_primary", "rays_bounce", "rays_shadow"):
                    assert getattr(got["stats"], k) == getattr(ref["stats"], k), k
                retraced.append(int(got["stats"].retraced))
            # the first frame has nothing remembered and is sized from the bound; the spoiled memory shows from the second on
            assert retraced == ([0, 1, 1] if spoil == "-9000" else [0, 0, 0]), retraced
            first = got["rgb"].copy()
            for _ in range(3):                                  # two interleaved shards, each with sizes of its own
                img = np.zeros((H, W, 3), np.float32)
                rays = sum(ex.render(W, H, mode=CPU, max_depth=depth, tone=False, tile_first=r, tile_step=2, out=img,
                                     want_ids=False)["stats"].rays for r in range(2))
                assert np.array_equal(img.view(np.int32), first.view(np.int32)) and rays == ref["stats"].rays
            for _ in range(3):                                  # asynchronous, device-resident: collected by sync()
                st = _abi.Stats()
                hip.memset(d_out, 0, ((H + 7) // 8) * 8 * W * 16)
                hip.sync()
                ex.render_device(W, H, d_out, None, mode=CPU, max_depth=depth, tone=False, stats=st)
                ex.sync()
                out = np.zeros((((H + 7) // 8) * 8, W, 4), np.float32)
                hip.d2h(out, d_out)
                assert np.array_equal(out[:H, :, :3].view(np.int32), first.view(np.int32))
                assert st.rays == ref["stats"].rays
    finally:
        hip.free(d_out)


@pytest.mark.parametrize("in_flight", [2, 3])
def test_frames_in_flight_equal_single_frames(in_flight):
    """Throughput mode: executors of one scene, one lane each, take asynchronous device-resident frames in turn.  Every
    frame is the oracle's (ids exact through the colours' parity, linear colour within 4 ULP), whatever is in flight
    beside it: frames of two different sizes and depths alternate."""
    from raytracer_amd.executor import FramesInFlight
    from oracle.binding import Oracle
    from tests.util import HipMem
    fs = scenegen.small_variant("mirror").to_flat()
    cfgs = [(48, 132, 4), (64, 40, 1)]
    refs = [Oracle(fs).render(W, H, mode=CPU, max_depth=d, tone=0, n_threads=8) for (W, H, d) in cfgs]
    hip = HipMem()
    size = max(((H + 7) // 8) * 8 * W * 16 for (W, H, _) in cfgs)
    bufs = [hip.malloc(size) for _ in range(in_flight)]
    try:
        with FramesInFlight(fs, device=0, depth=in_flight) as pipe:
            pending = []                                     # (slot, cfg index)
            done = 0

            def check(slot, ci):
                W, H, d = cfgs[ci]
                st = pipe.collect(slot)
                out = np.zeros((((H + 7) // 8) * 8, W, 4), np.float32)
                hip.d2h(out, bufs[slot])
                assert ulp_diff(out[:H, :, :3], refs[ci]["linear"]).max() <= 4
                assert (out[:H, :, 3] == 1).all()
                assert st.rays == refs[ci]["stats"].rays and st.lanes == 1
            for i in range(9):
                ci = i % 2
                W, H, d = cfgs[ci]
                if len(pending) == in_flight:
                    check(*pending.pop(0)); done += 1
                k = pipe.next
                hip.memset(bufs[k], 0, size)
                hip.sync()
                pending.append((pipe.submit(W, H, bufs[k], None, mode=CPU, max_depth=d, tone=False), ci))
            while pending:
                check(*pending.pop(0)); done += 1
            assert done == 9
    finally:
        for b in bufs:
            hip.free(b)


@pytest.mark.parametrize("tail_max", ["0", "300000"])
def test_refraction_mode_matches_its_cpu_definition(monkeypatch, tail_max):
    """The optional refraction mode (no reference counterpart; the reference only declares the two Material fields,
    scene.h:28-29) against the oracle's definition: glass spheres and a glass pane, rays entering, leaving and totally
    reflected inside; both algorithms, level-by-level and tail-kernel paths; off again afterwards."""
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    monkeypatch.setenv("HIPEXEC_TAIL_MAX", tail_max)
    fs = scenegen.small_variant("glass").to_flat()
    orc = Oracle(fs)
    W, H, depth = 72, 56, 5
    ref = orc.render(W, H, mode=CPU, max_depth=depth, tone=0, n_threads=8, refract=True)
    plain = orc.render(W, H, mode=CPU, max_depth=depth, tone=0, n_threads=8)
    assert not np.array_equal(ref["linear"], plain["linear"]) and ref["stats"].rays_bounce != plain["stats"].rays_bounce
    with HipExecutor(fs, device=0) as ex:
        for algo in ("bvh", "scan"):
            for _ in range(2):
                got = ex.render(W, H, mode=CPU, max_depth=depth, tone=False, refract=True, algo=algo)
                assert np.array_equal(got["ids"], ref["ids"])
                assert ulp_diff(got["rgb"], ref["linear"]).max() <= 4
                for k in ("rays_primary", "rays_bounce", "rays_shadow"):
                    assert getattr(got["stats"], k) == getattr(ref["stats"], k), (algo, k)
            back = ex.render(W, H, mode=CPU, max_depth=depth, tone=False, algo=algo)
            assert ulp_diff(back["rgb"], plain["linear"]).max() <= 4
            assert back["stats"].rays_bounce == plain["stats"].rays_bounce
        toned = ex.render(W, H, mode=CPU, max_depth=depth, refract=True)
        assert_image_parity(toned["rgb"], orc.render(W, H, mode=CPU, max_depth=depth, tone=1, n_threads=8, refract=True)["rgb"])


@pytest.mark.parametrize("name,depth", [("c1", 0), ("mirror", 2), ("c5", 0)])
def test_ambient_occlusion_mode_matches_its_cpu_definition(name, depth):
    """The reference's optional ambient occlusion (ENABLE_AO, ray_tracer.cpp:167-169,406-435) with its rand() replaced by a
    counter-based generator keyed on the hit point: 30 occlusion rays per hit, all levels, against the oracle's
    definition (ids exact, colours within 4 ULP, the occlusion rays counted as shadow rays); off again afterwards; the
    index-order scans refuse the mode."""
    from raytracer_amd.executor import HipExecutor, HipExecutorError
    from oracle.binding import Oracle
    fs = scenegen.small_variant(name).to_flat()
    fs.ambient_factor = 0.6
    orc = Oracle(fs)
    W, H = 56, 40
    ref = orc.render(W, H, mode=CPU, max_depth=depth, tone=0, n_threads=8, ao=True)
    plain = orc.render(W, H, mode=CPU, max_depth=depth, tone=0, n_threads=8)
    assert not np.array_equal(ref["linear"], plain["linear"])
    assert ref["stats"].rays_shadow >= plain["stats"].rays_shadow + 30 * int((ref["ids"] >= 0).sum())
    with HipExecutor(fs, device=0) as ex:
        for _ in range(2):
            got = ex.render(W, H, mode=CPU, max_depth=depth, tone=False, ao=True)
            assert np.array_equal(got["ids"], ref["ids"])
            assert ulp_diff(got["rgb"], ref["linear"]).max() <= 4
            for k in ("rays_primary", "rays_bounce", "rays_shadow"):
                assert getattr(got["stats"], k) == getattr(ref["stats"], k), k
        back = ex.render(W, H, mode=CPU, max_depth=depth, tone=False)
        assert ulp_diff(back["rgb"], plain["linear"]).max() <= 4 and back["stats"].rays_shadow == plain["stats"].rays_shadow
        with pytest.raises(HipExecutorError, match="hierarchy"):
            ex.render(W, H, mode=CPU, max_depth=depth, ao=True, algo="scan")
        again = ex.render(W, H, mode=CPU, max_depth=depth, tone=False)         # the refused call leaves the executor usable
        assert np.array_equal(again["rgb"].view(np.int32), back["rgb"].view(np.int32))


def test_executors_made_from_pooled_memory_render_the_same_frames():
    """hipexec_destroy hands device memory, pinned memory and streams to a pool that serves the next hipexec_create (the
    reference makes an executor per frame, ray_tracer_cl.cpp:14).  Executors of different scenes made one after the
    other therefore get each other's blocks with their old contents: every frame must still be the oracle's, for both
    algorithms, and hipexec_trim must leave the library usable."""
    from raytracer_amd import executor
    from raytracer_amd.executor import HipExecutor
    from oracle.binding import Oracle
    W, H = 64, 48
    scenes = [scenegen.small_variant(n).to_flat() for n in ("mirror", "soup", "c5", "mirror", "c2")]
    refs = [Oracle(fs).render(W, H, mode=CPU, max_depth=2, tone=0, n_threads=8) for fs in scenes]
    for round_ in range(3):
        for fs, ref in zip(scenes, refs):
            with HipExecutor(fs, device=0) as ex:                      # created, used once, destroyed
                for algo in ("bvh", "scan") if round_ == 1 else ("bvh",):
                    got = ex.render(W, H, mode=CPU, max_depth=2, tone=False, algo=algo)
                    assert np.array_equal(got["ids"], ref["ids"]), (round_, algo)
                    assert ulp_diff(got["rgb"], ref["linear"]).max() <= 4, (round_, algo)
                    assert got["stats"].rays == ref["stats"].rays
        if round_ == 1:
            executor.trim()                                            # everything cached goes back to the driver
    executor.trim(0)


@pytest.mark.parametrize("hard", [(12.5, 50.0, 7.25), (0.0, 1.0, 1024.0), (1500.0, 1025.0, 3.0), (0.5, 0.5, 0.5)])
@pytest.mark.parametrize("algo", ["bvh", "scan"])
def test_specular_exponents_whole_fractional_and_large(algo, hard):
    """powf(sd, hardness) of the Phong term (ray_tracer.cpp:396-401, ray_tracer_cl.cpp:166-171): whole-number exponents up to
    1024 take a multiplication chain on the device, everything else the generic evaluation, and the kernels that shade
    after tracing carry two copies of their lamp loop, chosen per wave by whether all of its samples have whole-number
    exponents.  Materials with mixed exponents put both kinds into one wave; each combination against the oracle's frame."""
    from oracle.binding import Oracle
    from raytracer_amd.executor import HipExecutor
    fs = scenegen.small_variant("c2").to_flat()
    fs.mat_hard = [float(hard[i % 3]) for i in range(len(fs.mat_hard))]
    fs.mat_ks = [0.9 for _ in fs.mat_ks]
    W, H, depth = 64, 56, 2
    orc = Oracle(fs)
    with HipExecutor(fs, device=0) as ex:
        for mode in (CPU, CL):
            ref = orc.render(W, H, mode=mode, max_depth=depth, tone=0, n_threads=8)
            got = ex.render(W, H, mode=mode, max_depth=depth, tone=False, algo=algo)
            assert np.array_equal(got["ids"], ref["ids"])
            assert ulp_diff(got["rgb"], ref["linear"]).max() <= 4, (mode, ulp_diff(got["rgb"], ref["linear"]).max())
