"""Committed regression vectors (tests/golden, written by make_golden.py from the CPU oracle):
the oracle must still reproduce them on this machine (CPU test), and the HIP path must match
them through the C ABI (GPU test).  They are oracle outputs, not reference outputs."""
import glob
import os
import re

import numpy as np
import pytest

from raytracer_amd import _abi, scenegen
from oracle import binding as orc
from tests.util import assert_image_parity, ulp_diff

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FRAMES = sorted(glob.glob(os.path.join(GOLD, "frame_*.npz")))


def _parse(path):
    m = re.match(r"frame_(.+)_d(\d+)_m(\d)\.npz", os.path.basename(path))
    return m.group(1), int(m.group(2)), int(m.group(3))


def test_predicate_vectors_oracle():
    g = np.load(os.path.join(GOLD, "predicates.npz"))
    n = g["tri_p"].shape[0]
    idx = np.arange(0, n, 3)                        # a third of the vectors keeps the CPU suite quick
    for mode, key in ((0, "tri_cpu"), (1, "tri_cl")):
        exp = g[key]
        for i in idx:
            hit, t, u, v, _ = orc.triangle_hit(mode, g["tri_p"][i], g["tri_e1"][i], g["tri_e2"][i], g["tri_o"][i], g["tri_d"][i])
            got = np.array((hit, t, u, v) if (hit or mode == 1) else (0, 0, 0, 0), np.float32)
            assert np.array_equal(got.view(np.int32), exp[i].view(np.int32)), (mode, i)
    for mode, key in ((0, "sph_cpu"), (1, "sph_cl")):
        exp = g[key]
        for i in idx:
            hit, t, _, _ = orc.sphere_hit(mode, g["sph_c"][i], g["sph_r"][i], g["sph_o"][i], g["sph_d"][i])
            got = np.array((hit, t if hit else 0), np.float32)
            assert np.array_equal(got.view(np.int32), exp[i].view(np.int32)), (mode, i)
    # the vectors exercise both outcomes and the edge cases
    assert 0.2 < g["tri_cpu"][:, 0].mean() < 0.8 and g["tri_cl"][:, 0].mean() < g["tri_cpu"][:, 0].mean()
    assert 0.2 < g["sph_cpu"][:, 0].mean() < 0.9


@pytest.mark.parametrize("path", FRAMES, ids=[os.path.basename(p) for p in FRAMES])
def test_frames_oracle(path):
    name, depth, mode = _parse(path)
    g = np.load(path)
    H, W = g["ids"].shape[:2]
    r = orc.Oracle(scenegen.small_variant(name).to_flat()).render(W, H, mode=mode, max_depth=depth, tone=1, n_threads=4)
    assert np.array_equal(r["ids"], g["ids"])
    assert np.array_equal(r["linear"].view(np.int32), g["linear"].view(np.int32))
    assert ulp_diff(r["rgb"], g["rgb"]).max() <= 1          # powf of the host libm
    st = r["stats"]
    assert list(g["counts"]) == [st.rays_primary, st.rays_bounce, st.rays_shadow, st.tests_tri_closest,
                                 st.tests_sph_closest, st.tests_tri_any, st.tests_sph_any]


@pytest.mark.gpu
@pytest.mark.parametrize("path", FRAMES, ids=[os.path.basename(p) for p in FRAMES])
def test_frames_gpu(path):
    from raytracer_amd.executor import HipExecutor
    name, depth, mode = _parse(path)
    g = np.load(path)
    H, W = g["ids"].shape[:2]
    with HipExecutor(scenegen.small_variant(name).to_flat(), device=0) as ex:
        for algo in ("bvh", "scan"):
            r = ex.render(W, H, mode=mode, max_depth=depth, tone=True, algo=algo)
            lin = ex.render(W, H, mode=mode, max_depth=depth, tone=False, want_ids=False, algo=algo)
            assert np.array_equal(r["ids"], g["ids"]), algo
            assert ulp_diff(lin["rgb"], g["linear"]).max() <= 4
            assert_image_parity(r["rgb"], g["rgb"])
            st = r["stats"]
            n_cmp = 7 if algo == "scan" else 5        # the reference's any-hit test counts need index-order scans
            assert list(g["counts"])[:n_cmp] == [st.rays_primary, st.rays_bounce, st.rays_shadow, st.tests_tri_closest,
                                                 st.tests_sph_closest, st.tests_tri_any, st.tests_sph_any][:n_cmp]


@pytest.mark.gpu
def test_predicate_vectors_gpu():
    """Single-triangle / single-sphere scenes through hipexec_closest_hits reproduce the predicate vectors."""
    from raytracer_amd.executor import HipExecutor
    from raytracer_amd.scene import FlatScene
    g = np.load(os.path.join(GOLD, "predicates.npz"))
    # group the vectors by primitive: one scene holds a batch of triangles, ray i is only compared with primitive i
    n = 400
    fs = FlatScene()
    fs.tri_p, fs.tri_e1, fs.tri_e2 = g["tri_p"][:n], g["tri_e1"][:n], g["tri_e2"][:n]
    fs.tri_norm = np.zeros((n, 3), np.float32); fs.tri_mat = np.full(n, -1, np.int32)
    fs.sph_center, fs.sph_radius, fs.sph_mat = g["sph_c"][:n], g["sph_r"][:n], np.full(n, -1, np.int32)
    fs = fs.normalised()
    o = orc.Oracle(fs)
    rays_t = np.concatenate([g["tri_o"][:n], g["tri_d"][:n]], axis=1)
    rays_s = np.concatenate([g["sph_o"][:n], g["sph_d"][:n]], axis=1)
    with HipExecutor(fs, device=0) as ex:
        for mode in (0, 1):
            for rays in (rays_t, rays_s):
                a, b = ex.closest_hits(rays, mode), o.closest_hits(rays, mode)
                assert np.array_equal(a["id"], b["id"])
                for f in ("t", "u", "v", "point", "norm"):
                    assert np.array_equal(a[f].view(np.int32), b[f].view(np.int32)), f
                assert np.array_equal(ex.any_hits(rays, mode) != 0, o.any_hits(rays, mode) != 0)
