"""The C++ scene loader and headless CLI (integration/scene_loader.hpp, render_cli.cpp).

CPU part: the loader's flattened arrays equal the Python loader's (raytracer_amd/scene.py,
itself checked against hand-computed values in test_scene_loader.py) bit for bit on scenes that
exercise textures, null materials, faces without UVs, missing world block and fractional
sphere / lamp coordinates.  GPU part: a frame rendered by the CLI matches the oracle's frame of the
same scene file (hit ids exact, RGB8 gate) and equals the frame rendered through the Python binding."""
import os
import struct
import subprocess

import numpy as np
import pytest

from raytracer_amd import scenegen
from raytracer_amd.scene import load_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "raytracer_amd", "bin", "hiprt_render")

DTYPES = {"tri_mat": np.int32, "sph_mat": np.int32, "mat_tex": np.int32, "tex_rgba": np.uint8,
          "tex_offset": np.uint64, "tex_w": np.uint32, "tex_h": np.uint32}


def read_dump(path):
    out = {}
    with open(path, "rb") as fh:
        while True:
            tag = fh.read(16)
            if not tag:
                break
            (n,) = struct.unpack("<Q", fh.read(8))
            name = tag.rstrip(b"\0").decode()
            out[name] = np.frombuffer(fh.read(n), dtype=DTYPES.get(name, np.float32))
    return out


def needs_cli():
    if not os.path.exists(CLI):
        pytest.fail("raytracer_amd/bin/hiprt_render is missing: run __graft_entry__.build()")


@pytest.mark.parametrize("name", ["ties", "c2", "c1_frac", "c4", "soup"])
def test_cpp_loader_matches_python_loader(tmp_path, name):
    needs_cli()
    path = scenegen.small_variant(name).write(str(tmp_path / f"{name}.scene"))
    ref = load_scene(path)
    dump = str(tmp_path / "scene.bin")
    subprocess.run([CLI, path, "--dump-scene", dump], check=True, capture_output=True)
    got = read_dump(dump)
    for k in ("tri_p", "tri_e1", "tri_e2", "tri_norm", "tri_uv", "sph_center", "sph_radius", "mat_color", "mat_kd",
              "mat_ks", "mat_hard", "mat_kr", "lamp_pos", "lamp_intensity", "lamp_distance", "cam_pos", "cam_mat",
              "horizon", "ambient"):
        a = np.ascontiguousarray(getattr(ref, k), dtype=np.float32).reshape(-1)
        assert np.array_equal(a.view(np.int32), got[k].view(np.int32)), k
    for k in ("tri_mat", "sph_mat", "mat_tex"):
        assert np.array_equal(np.asarray(getattr(ref, k), np.int32).reshape(-1), got[k]), k
    assert list(got["tex_w"]) == [t.shape[1] for t in ref.textures]
    assert list(got["tex_h"]) == [t.shape[0] for t in ref.textures]
    atlas = np.concatenate([t.reshape(-1) for t in ref.textures]) if ref.textures else np.zeros((0,), np.uint8)
    assert np.array_equal(atlas, got["tex_rgba"])


def test_cpp_loader_reports_errors(tmp_path):
    needs_cli()
    bad = tmp_path / "bad.scene"
    bad.write_text('{"world": null, "materials": [], "vertices": [[0,0,0]], "faces": [{"vertices": [0,1,2], "material": null, "uv": null}],'
                   ' "spheres": [], "lamps": [], "transform": [[1,0,0],[0,1,0],[0,0,1]], "translate": [0,0,0]}')
    r = subprocess.run([CLI, str(bad), "--dump-scene", str(tmp_path / "x.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "out of range" in r.stderr
    r = subprocess.run([CLI, str(tmp_path / "missing.scene")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr
    bad.write_text('{"world": null,')
    r = subprocess.run([CLI, str(bad), "--dump-scene", str(tmp_path / "x.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "json" in r.stderr


def test_png_decoder_colour_types(tmp_path):
    """Grey, grey+alpha, RGB, palette and 16-bit PNGs decode to the RGBA8 PIL gives."""
    needs_cli()
    from PIL import Image
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (12, 20, 4), dtype=np.uint8)
    images = {
        "rgba": Image.fromarray(base, "RGBA"),
        "rgb": Image.fromarray(base[..., :3].copy(), "RGB"),
        "grey": Image.fromarray(base[..., 0].copy(), "L"),
        "la": Image.fromarray(base[..., :2].copy(), "LA"),
        "pal": Image.fromarray(base[..., :3].copy(), "RGB").quantize(16),
    }
    for key, im in images.items():
        p = str(tmp_path / f"{key}.png")
        im.save(p)
        spec = scenegen.small_variant("ties")
        doc = spec.to_json_doc(texture_dir=str(tmp_path / "t"))
        doc["materials"][0]["imagePath"] = p
        import json
        sp = str(tmp_path / f"{key}.scene")
        json.dump(doc, open(sp, "w"))
        dump = str(tmp_path / f"{key}.bin")
        subprocess.run([CLI, sp, "--dump-scene", dump], check=True, capture_output=True)
        got = read_dump(dump)
        want = np.asarray(Image.open(p).convert("RGBA"), dtype=np.uint8).reshape(-1)
        assert np.array_equal(got["tex_rgba"], want), key


@pytest.mark.gpu
@pytest.mark.parametrize("mode,depth", [("cl", 0), ("cpu", 2)])
def test_cli_frame_matches_oracle_and_binding(tmp_path, mode, depth):
    needs_cli()
    from raytracer_amd import _abi
    from raytracer_amd.executor import HipExecutor
    path = scenegen.small_variant("c2").write(str(tmp_path / "c2.scene"))
    W, H = 72, 40
    pfm, ids = str(tmp_path / "o.pfm"), str(tmp_path / "ids.bin")
    r = subprocess.run([CLI, path, "-o", pfm, "--ids", ids, "--width", str(W), "--height", str(H), "--mode", mode,
                        "--depth", str(depth), "--frames", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("Mrays/s") == 2
    raw = open(pfm, "rb").read()
    head = f"PF\n{W} {H}\n-1.0\n".encode()
    assert raw.startswith(head)
    img = np.frombuffer(raw[len(head):], np.float32).reshape(H, W, 3)
    m = _abi.HIPEXEC_MODE_CL if mode == "cl" else _abi.HIPEXEC_MODE_CPU
    # the CLI's frame (C++ loader -> C ABI -> file) against the oracle on the scene the Python loader reads
    from oracle.binding import Oracle
    from tests.util import assert_image_parity
    oref = Oracle(load_scene(path)).render(W, H, mode=m, max_depth=depth, tone=1, n_threads=8)
    assert np.array_equal(np.fromfile(ids, np.int32).reshape(oref["ids"].shape), oref["ids"])
    assert_image_parity(img, oref["rgb"], min_frac=1.0)
    with HipExecutor(load_scene(path), device=0) as ex:
        ref = ex.render(W, H, mode=m, max_depth=depth, tone=True)
    assert np.array_equal(img.view(np.int32), ref["rgb"].view(np.int32))
    got_ids = np.fromfile(ids, np.int32).reshape(ref["ids"].shape)
    assert np.array_equal(got_ids, ref["ids"])
    # --optimize (hipexec_optimize before the first frame): the same file, bit for bit
    pfm2 = str(tmp_path / "o2.pfm")
    r = subprocess.run([CLI, path, "-o", pfm2, "--width", str(W), "--height", str(H), "--mode", mode, "--depth", str(depth),
                        "--optimize"], capture_output=True, text=True)
    assert r.returncode == 0 and "surface-area topology" in r.stderr, r.stderr
    assert open(pfm2, "rb").read() == raw
    png = str(tmp_path / "o.png")
    subprocess.run([CLI, path, "-o", png, "--width", str(W), "--height", str(H), "--mode", mode, "--depth", str(depth)],
                   check=True, capture_output=True)
    from PIL import Image
    px = np.asarray(Image.open(png).convert("RGB"), dtype=np.uint8)
    want = np.floor(np.clip(ref["rgb"][::-1], 0, 1) * np.float32(255) + np.float32(0.5)).astype(np.uint8)
    assert np.array_equal(px, want)
