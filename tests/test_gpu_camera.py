"""Executors that share a scene (hipexec_clone) and cameras that change (hipexec_set_camera), against the oracle.

The reference takes the camera from the Scene it is handed with every call (ray_tracer_cl.cpp:80-92) and builds an
executor per frame (:14); here an executor keeps the scene on the device, so a new camera is a call of its own, and
several executors of one scene share the device copy.  Every frame below is compared with the oracle's frame of the
same scene arrays and that camera: hit ids exact, linear colour within 4 ULP, ray counts equal.

The launches of the recursion levels are sized on the host, from the sizes the last frame of the same configuration
produced plus a margin, or from the bound n(L+1) <= n(L); a level that turns out larger
than its launch makes the executor trace the frame again.  The camera sequences below go through all of those cases."""
import copy

import numpy as np
import pytest

from raytracer_amd import _abi, scenegen
from tests.util import HipMem, ulp_diff

pytestmark = pytest.mark.gpu

CPU, CL = _abi.HIPEXEC_MODE_CPU, _abi.HIPEXEC_MODE_CL


def _rot_y(a):
    c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)


def _rot_x(a):
    c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float32)


def _with_camera(fs, pos, mat):
    g = copy.copy(fs)
    g.cam_pos = np.asarray(pos, np.float32).copy()
    g.cam_mat = np.asarray(mat, np.float32).reshape(3, 3).copy()
    return g


def _cameras(fs):
    """The scene's own camera, small steps away from it, a look at the sky (no bounce rays) held for two frames, back at
    the mirror spheres from close by (far more than a launch sized from the sky frames covers), and one from outside the
    scene's bounds."""
    p0, m0 = fs.cam_pos.copy(), fs.cam_mat.copy()
    cams = [(p0, m0)]
    for k in range(1, 4):
        cams.append((p0 + np.float32(0.15 * k) * np.array([1, 0.2, -0.5], np.float32), (m0 @ _rot_y(0.03 * k)).astype(np.float32)))
    cams.append((p0, (m0 @ _rot_x(1.2)).astype(np.float32)))                        # up at the sky
    cams.append(cams[-1])                                                           # and again: its sizes now count as settled
    cams.append((p0 + np.array([0, -0.8, -4.0], np.float32), m0))                   # close to the mirrors
    cams.append((p0 * np.float32(40.0), m0))                                        # far outside the scene
    cams.append((p0, m0))
    return cams


def _check(got, ref, what):
    assert np.array_equal(got["ids"], ref["ids"]), what
    assert ulp_diff(got["rgb"], ref["linear"]).max() <= 4, what
    for k in ("rays_primary", "rays_bounce", "rays_shadow", "tests_tri_closest", "tests_sph_closest"):
        assert getattr(got["stats"], k) == getattr(ref["stats"], k), (what, k)


@pytest.mark.parametrize("algo", ["bvh", "scan"])
def test_camera_changes_give_the_oracles_frames(algo):
    from oracle.binding import Oracle
    from raytracer_amd.executor import HipExecutor
    fs = scenegen.small_variant("c2").to_flat()
    W, H, depth = 48, 132, 4
    bounce, retraced = [], []
    with HipExecutor(fs, device=0) as ex:
        ex.set_algorithm(algo)
        for i, (pos, mat) in enumerate(_cameras(fs)):
            ex.set_camera(pos, mat)
            ref = Oracle(_with_camera(fs, pos, mat)).render(W, H, mode=CPU, max_depth=depth, tone=0, n_threads=8)
            got = ex.render(W, H, mode=CPU, max_depth=depth, tone=False)
            _check(got, ref, f"camera {i}")
            retraced.append(int(got["stats"].retraced))
            bounce.append(int(ref["stats"].rays_bounce))
            if i in (0, 6):                                                  # the same camera again: same frame
                _check(ex.render(W, H, mode=CPU, max_depth=depth, tone=False), ref, f"camera {i} again")
            refc = Oracle(_with_camera(fs, pos, mat)).render(W, H, mode=CL, tone=0, n_threads=8)
            gotc = ex.render(W, H, mode=CL, tone=False)
            assert np.array_equal(gotc["ids"], refc["ids"]) and ulp_diff(gotc["rgb"], refc["linear"]).max() <= 4
    # the sequence really has a frame whose queues outgrow a launch sized from the (settled) frames before it, frames that
    # shrink, and frames that change a little; the hierarchy algorithm traced that one frame twice and no other (small
    # steps stay inside the margin: a quarter, twice the last change and 4096 rays), the scans wait for every size
    assert bounce[6] > 2 * (bounce[5] + bounce[5] // 4 + 4096) and bounce[5] < bounce[0] // 2 and bounce[1] != bounce[0]
    assert (retraced[:7] == [0, 0, 0, 0, 0, 0, 1]) if algo == "bvh" else (retraced == [0] * 9), retraced


def test_a_frame_that_outgrows_its_launches_is_traced_again(monkeypatch):
    """Remembered sizes that are too small (the test hook stores them 30 % below what the kernels produced): the frame
    sized from them loses rays on the device, the check finds out, and the caller still gets the oracle's frame."""
    from oracle.binding import Oracle
    from raytracer_amd.executor import HipExecutor
    fs = scenegen.small_variant("mirror").to_flat()
    W, H, depth = 48, 132, 4
    ref = Oracle(fs).render(W, H, mode=CPU, max_depth=depth, tone=0, n_threads=8)
    monkeypatch.setenv("HIPEXEC_PREDICT_SPOIL", str(-int(0.3 * ref["stats"].rays_bounce / 2)))
    with HipExecutor(fs, device=0) as ex:
        ex.set_lanes(1)
        for i in range(4):
            _check(ex.render(W, H, mode=CPU, max_depth=depth, tone=False), ref, f"frame {i}")


def test_clones_share_the_scene_and_keep_their_own_cameras():
    """Three executors of one uploaded scene (hipexec_clone), each with a camera of its own, frames submitted
    asynchronously on all of them before any is collected; the source is destroyed first."""
    from oracle.binding import Oracle
    from raytracer_amd.executor import HipExecutor
    fs = scenegen.small_variant("c2").to_flat()
    W, H, depth = 64, 72, 3
    cams = _cameras(fs)[:3]
    refs = [Oracle(_with_camera(fs, p, m)).render(W, H, mode=CPU, max_depth=depth, tone=0, n_threads=8) for p, m in cams]
    hip = HipMem()
    rows = ((H + 7) // 8) * 8
    bufs = [hip.malloc(rows * W * 16) for _ in cams]
    src = HipExecutor(fs, device=0)
    exs = [src.clone() for _ in cams]
    try:
        for e, (p, m) in zip(exs, cams):
            e.set_lanes(1)
            e.set_camera(p, m)
        src.close()                                                          # the clones keep the scene alive
        for round_ in range(3):
            stats = [_abi.Stats() for _ in cams]
            for e, b, st in zip(exs, bufs, stats):
                e.render_device(W, H, b, None, mode=CPU, max_depth=depth, tone=False, stats=st)
            for e, b, st, ref in zip(exs, bufs, stats, refs):
                e.sync()
                out = np.zeros((rows, W, 4), np.float32)
                hip.d2h(out, b)
                assert ulp_diff(out[:H, :, :3], ref["linear"]).max() <= 4, round_
                assert st.rays == ref["stats"].rays
        # the scans of a clone: its own camera records next to the shared ones
        for e, ref in zip(exs, refs):
            got = e.render(W, H, mode=CPU, max_depth=depth, tone=False, algo="scan")
            assert np.array_equal(got["ids"], ref["ids"]) and ulp_diff(got["rgb"], ref["linear"]).max() <= 4
    finally:
        for e in exs:
            e.close()
        for b in bufs:
            hip.free(b)


def test_clone_of_a_large_texture_set_takes_no_second_copy():
    """Eight executors of a scene with 64 MiB of textures hold one copy of it: after the first executor exists, the
    clones may take work buffers and streams, not another atlas."""
    from raytracer_amd.executor import HipExecutor, trim
    fs = scenegen.small_variant("c4").to_flat()
    rng = np.random.default_rng(4)
    fs.textures = [rng.integers(0, 256, size=(2048, 2048, 4), dtype=np.uint8) for _ in range(len(fs.textures))]
    atlas = sum(t.size for t in fs.textures)
    assert atlas >= 16 << 20
    hip = HipMem()
    trim()
    with HipExecutor(fs, device=0) as src:
        src.render(32, 24, want_ids=False)
        hip.sync()
        before = hip.mem_free()
        clones = [src.clone() for _ in range(7)]
        try:
            for e in clones:
                e.render(32, 24, want_ids=False)
            hip.sync()
            # (seven private copies would be 7 x 64 MiB; what the clones do take is a few allocation granules each
            # for their work buffers and counters)
            assert before - hip.mem_free() < 2 * atlas, "the clones took memory of the order of seven texture atlases"
        finally:
            for e in clones:
                e.close()


def test_levels_left_to_the_tail_kernel_are_not_remembered_as_empty(monkeypatch):
    """k_bvh_tail takes a batch over from the first level whose launch is small enough, and the levels after it are then
    not counted one by one.  Whether a level is small enough depends on how its launch was sized: with the exact
    remembered size level 1 is (the knob puts the limit at exactly n(1)), with the margin of a frame that follows a
    change it is not, and level 2 then gets a launch of its own -- which must be sized from what is known of level 2
    (an older frame's count, or the bound), not from the zero the tail-kernel frame left.  No frame of the static
    sequence may be traced twice, and every one is the oracle's."""
    from oracle.binding import Oracle
    from raytracer_amd.executor import HipExecutor
    fs = scenegen.small_variant("c2").to_flat()
    W, H, depth = 192, 264, 3
    orc = Oracle(fs)
    ref = orc.render(W, H, mode=CPU, max_depth=depth, tone=0, n_threads=8)
    n1 = int(orc.render(W, H, mode=CPU, max_depth=1, tone=0, n_threads=8)["stats"].rays_bounce)
    n2 = int(orc.render(W, H, mode=CPU, max_depth=2, tone=0, n_threads=8)["stats"].rays_bounce) - n1
    n0 = W * H * 4
    # level 1 sized with the margin (capped by the bound n0) is above the limit; level 2 is more than the margin's slack
    # alone covers, and small enough for the tail kernel either way
    assert n1 < n0 and n2 > 2 * 4096 and n2 + n2 // 4 + 4096 <= n1, (n0, n1, n2)
    monkeypatch.setenv("HIPEXEC_KNOBS", "1")
    monkeypatch.setenv("HIPEXEC_TAIL_MAX", str(n1))
    retraced = []
    with HipExecutor(fs, device=0) as ex:
        ex.set_lanes(1)
        for i in range(8):
            got = ex.render(W, H, mode=CPU, max_depth=depth, tone=False)
            _check(got, ref, f"frame {i}")
            retraced.append(int(got["stats"].retraced))
    assert retraced == [0] * 8, retraced
