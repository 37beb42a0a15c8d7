"""The C++ drop-in (integration/ray_tracer_hip.cpp + scene_flatten.hpp), compiled against a
test stand-in for the reference's types (tests/cpp/mock_reference.h; the reference's scene.h
needs glm, which this image lacks).  CPU: it compiles, also against the two headers of the
reference that stand on their own (image_bitmap.h, config.h) where the reference tree exists,
and the Scene -> SoA flattening reproduces the arrays.  GPU: the five ray_tracer_cl.h functions
give the CL-mode results of the oracle."""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import _abi, scenegen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


REFERENCE = "/root/reference"


def _build(out, extra=()):
    libdir = os.path.join(ROOT, "raytracer_amd", "lib")
    cmd = ["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-ffp-contract=off",
           '-DRT_HIP_REFERENCE_HEADER="mock_reference.h"', *extra,
           "-I", os.path.join(ROOT, "tests", "cpp"), "-I", os.path.join(ROOT, "integration"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "dropin_driver.cpp"), os.path.join(ROOT, "integration", "ray_tracer_hip.cpp"),
           "-L", libdir, "-lhip_executor", f"-Wl,-rpath,{libdir}", "-o", out]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dropin") / "dropin_driver")
    r = _build(out)
    assert r.returncode == 0, r.stderr
    return out


def _dump(fs, path):
    with open(path, "wb") as f:
        np.array([fs.n_tri, fs.n_sph, fs.mat_kd.shape[0], len(fs.textures), fs.n_lamp], np.int32).tofile(f)
        for a in (fs.tri_p, fs.tri_e1, fs.tri_e2, fs.tri_norm, fs.tri_uv, fs.tri_mat, fs.sph_center, fs.sph_radius, fs.sph_mat,
                  fs.mat_color, fs.mat_kd, fs.mat_ks, fs.mat_hard, fs.mat_kr, fs.mat_tex):
            a.tofile(f)
        for t in fs.textures:
            np.array([t.shape[1], t.shape[0]], np.int32).tofile(f); t.tofile(f)
        for a in (fs.lamp_pos, fs.lamp_intensity, fs.lamp_distance, fs.cam_pos, fs.cam_mat, fs.horizon, fs.ambient):
            a.tofile(f)


def _load(path):
    raw = open(path, "rb").read()
    pos = [0]

    def take(dt, n):
        a = np.frombuffer(raw, dtype=dt, count=n, offset=pos[0]); pos[0] += a.nbytes
        return a
    nt, ns, nm, nx, nl = take(np.int32, 5)
    d = {"tri_p": take(np.float32, 3 * nt), "tri_e1": take(np.float32, 3 * nt), "tri_e2": take(np.float32, 3 * nt),
         "tri_norm": take(np.float32, 3 * nt), "tri_uv": take(np.float32, 6 * nt), "tri_mat": take(np.int32, nt),
         "sph_center": take(np.float32, 3 * ns), "sph_radius": take(np.float32, ns), "sph_mat": take(np.int32, ns),
         "mat_color": take(np.float32, 3 * nm).reshape(-1, 3), "mat_kd": take(np.float32, nm), "mat_ks": take(np.float32, nm),
         "mat_hard": take(np.float32, nm), "mat_kr": take(np.float32, nm), "mat_tex": take(np.int32, nm), "tex": []}
    for _ in range(nx):
        w, h = take(np.int32, 2)
        d["tex"].append(take(np.uint8, int(w) * int(h) * 4).reshape(h, w, 4))
    for k, n in (("lamp_pos", 3 * nl), ("lamp_intensity", nl), ("lamp_distance", nl), ("cam_pos", 3), ("cam_mat", 9),
                 ("horizon", 3), ("ambient", 3)):
        d[k] = take(np.float32, n)
    assert pos[0] == len(raw)
    return d


@pytest.mark.parametrize("name", ["ties", "c2", "c4"])
def test_flatten_reproduces_the_arrays(driver, tmp_path, name):
    fs = scenegen.small_variant(name).to_flat()
    src, dst = str(tmp_path / "scene.bin"), str(tmp_path / "flat.bin")
    _dump(fs, src)
    subprocess.run([driver, "flatten", src, dst], check=True)
    d = _load(dst)
    for k in ("tri_p", "tri_e1", "tri_e2", "tri_norm", "tri_uv", "sph_center", "sph_radius", "lamp_pos", "lamp_intensity",
              "lamp_distance", "cam_pos", "cam_mat", "horizon", "ambient"):
        assert np.array_equal(d[k].view(np.int32), getattr(fs, k).reshape(-1).view(np.int32)), k
    nm = fs.mat_kd.shape[0]
    assert np.array_equal(d["mat_color"][:nm], fs.mat_color) and np.array_equal(d["mat_kd"][:nm], fs.mat_kd)
    assert np.array_equal(d["mat_tex"][:nm], fs.mat_tex) and len(d["tex"]) == len(fs.textures)
    for a, b in zip(d["tex"], fs.textures):
        assert np.array_equal(a, b)
    # objects without a JSON material own a default Material: appended after the scene's materials, all zero
    for got, ref in ((d["tri_mat"], fs.tri_mat), (d["sph_mat"], fs.sph_mat)):
        assert np.array_equal(got[ref >= 0], ref[ref >= 0])
        extra = got[ref < 0]
        assert (extra >= nm).all()
        assert (d["mat_kd"][extra] == 0).all() and (d["mat_color"][extra] == 0).all() and (d["mat_tex"][extra] == -1).all()


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE, "image_bitmap.h")), reason="the reference tree only exists in the build container")
def test_dropin_compiles_against_the_references_standalone_headers(tmp_path):
    """image_bitmap.h and config.h need nothing but the standard library, so here the drop-in and its driver are compiled
    with the reference's OWN framebuffer class and block size in place of the stand-ins (-I/root/reference; nothing of
    the reference travels: the GPU box keeps the stand-in build).  A drift between what ray_tracer_hip.cpp calls
    (getWidth / getHeight / getRawData / setPixel) and what the reference's class offers fails this build."""
    out = str(tmp_path / "dropin_driver_ref")
    r = _build(out, ["-DRT_USE_REFERENCE_STANDALONE_HEADERS", "-I", REFERENCE])
    assert r.returncode == 0, r.stderr
    # ... and it is the reference's class that went in: the stand-in keeps its pixels in a std::vector named `d`,
    # the reference's in `mData` (image_bitmap.h:13); config.h's guard macro is defined in that build only
    probe = tmp_path / "probe.cpp"
    probe.write_text('#define RT_USE_REFERENCE_STANDALONE_HEADERS\n#include "mock_reference.h"\n'
                     "#ifndef RAY_TRACING_CONFIG_H\n#error config.h of the reference was not included\n#endif\n"
                     "#ifndef RAY_TRACING_IMAGEBITMAP_H\n#error image_bitmap.h of the reference was not included\n#endif\n"
                     "static_assert(SUB_BLOCK_WIDTH == 48 && SUB_BLOCK_HEIGHT == 48, \"block size\");\nint main() { image_bitmap b(3, 2); return b.getWidth() * 10 + b.getHeight() - 32; }\n")
    exe = str(tmp_path / "probe")
    subprocess.run(["g++", "-std=c++11", "-I", os.path.join(ROOT, "tests", "cpp"), "-I", REFERENCE, str(probe), "-o", exe], check=True)
    assert subprocess.run([exe]).returncode == 0
    # the flattening through that build gives the same arrays
    fs = scenegen.small_variant("c2").to_flat()
    src, dst = str(tmp_path / "scene.bin"), str(tmp_path / "flat.bin")
    _dump(fs, src)
    subprocess.run([out, "flatten", src, dst], check=True)
    d = _load(dst)
    assert np.array_equal(d["tri_p"].view(np.int32), fs.tri_p.reshape(-1).view(np.int32))
    assert np.array_equal(d["cam_mat"].view(np.int32), fs.cam_mat.reshape(-1).view(np.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ties", "c2"])
def test_dropin_functions_match_the_cl_mode_oracle(driver, tmp_path, name):
    from oracle.binding import Oracle
    from tests.util import assert_image_parity, ulp_diff
    fs = scenegen.small_variant(name).to_flat()
    W, H = 70, 52                                           # ragged against the 48x48 block walk
    rng = np.random.default_rng(3)
    o = rng.uniform([-8, 0.5, -14], [8, 8, 6], size=(500, 3)); d = rng.uniform([-10, -0.5, -18], [10, 3, 2], size=(500, 3)) - o
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    src, rp, dst = str(tmp_path / "scene.bin"), str(tmp_path / "rays.bin"), str(tmp_path / "out.bin")
    _dump(fs, src)
    with open(rp, "wb") as f:
        np.array([rays.shape[0]], np.int32).tofile(f); rays.tofile(f)
    subprocess.run([driver, "render", src, rp, dst, str(W), str(H)], check=True)
    raw = np.fromfile(dst, dtype=np.uint8)
    n = rays.shape[0]
    off = 0
    full = raw[off:off + W * H * 12].view(np.float32).reshape(H, W, 3); off += W * H * 12
    blocks = raw[off:off + W * H * 12].view(np.float32).reshape(H, W, 3); off += W * H * 12
    hits = raw[off:off + n * 48].view(np.float32).reshape(n, 12); off += n * 48
    occ = raw[off:off + n].view(np.int8); off += n
    cols = raw[off:off + n * 12].view(np.float32).reshape(n, 3); off += n * 12
    moved = raw[off:off + W * H * 12].view(np.float32).reshape(H, W, 3)
    orc = Oracle(fs)
    ref = orc.render(W, H, mode=_abi.HIPEXEC_MODE_CL, tone=1)
    assert_image_parity(full, ref["rgb"], min_frac=1.0)
    # the 48x48 block walk through renderSubBlockCl (ragged right, bottom and corner blocks): the pixels renderSceneCl gives
    assert np.array_equal(blocks.view(np.int32), full.view(np.int32))
    assert_image_parity(blocks, ref["rgb"], min_frac=1.0)
    h = orc.closest_hits(rays, _abi.HIPEXEC_MODE_CL)
    assert np.array_equal(hits[:, 0] > 0, h["id"] >= 0)
    m = h["id"] >= 0
    assert np.array_equal(hits[m, 1].view(np.int32), h["t"][m].view(np.int32))
    assert np.array_equal(hits[m, 2:5].view(np.int32), h["point"][m].view(np.int32))
    assert np.array_equal(hits[m, 5:8].view(np.int32), h["norm"][m].view(np.int32))
    assert np.array_equal(hits[m, 8:11].view(np.int32), h["color"][m].view(np.int32))
    kd = np.where(h["mat"] >= 0, fs.mat_kd[np.maximum(h["mat"], 0)], 0.0)
    assert np.array_equal(hits[m, 11], kd[m].astype(np.float32))
    pre = np.zeros(n, np.int8); pre[::5] = 1
    assert np.array_equal(occ != 0, orc.any_hits(rays, _abi.HIPEXEC_MODE_CL, hits=pre) != 0)
    base = np.tile(np.array([0.5, 0.25, 0.125], np.float32), (n, 1))
    assert ulp_diff(cols, orc.trace_rays(rays, _abi.HIPEXEC_MODE_CL, colours=base)).max() <= 4
    # renderSubBlockCl with a scene whose camera differs from the executor's: the frame of THAT camera
    import copy
    fs2 = copy.copy(fs)
    fs2.cam_pos = (fs.cam_pos + np.array([0.75, 0.5, -1.25], np.float32)).astype(np.float32)
    ref2 = Oracle(fs2).render(W, H, mode=_abi.HIPEXEC_MODE_CL, tone=1)
    assert_image_parity(moved, ref2["rgb"], min_frac=1.0)
    assert not np.array_equal(moved, full)
