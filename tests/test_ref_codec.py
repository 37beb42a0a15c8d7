"""File-format readers against the REFERENCE's own third-party code.

The reference decodes textures with lib/lodepng.cpp (tex_image.h:25-31) and parses scenes with lib/json.h
(scene.cpp:20-22, number conversions :53-57, :97-99).  Both compile unmodified in the build container
(`make -C oracle ref`); tests/golden/ref_codec/ holds input files and what those two libraries made of them
(generator: tests/golden/make_ref_codec_golden.py).  Here integration/png_min.hpp and json_min.hpp must give
the same bytes: RGBA8 pixels, accept/reject verdicts, float bit patterns and int truncations.  Where the
reference build exists, the committed fixtures are additionally re-derived from it."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "ref_codec")
REF_CODEC = os.path.join(ROOT, "oracle", "_ref", "ref_codec")


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("codec") / "codec_probe")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-ffp-contract=off", "-I", os.path.join(ROOT, "integration"),
                    os.path.join(ROOT, "tests", "cpp", "codec_probe.cpp"), "-lz", "-o", out], check=True)
    return out


@pytest.fixture(scope="module")
def expected():
    return dict(np.load(os.path.join(GOLD, "expected.npz")))


def _png_names():
    return sorted(f[:-4] for f in os.listdir(GOLD) if f.endswith(".png"))


def _decode(tool, path, tmp):
    subprocess.run([tool, "png", path, tmp], check=True, capture_output=True)
    raw = open(tmp, "rb").read()
    err, w, h = struct.unpack("<III", raw[:12])
    return err, w, h, np.frombuffer(raw[12:], np.uint8)


def test_fixture_set_covers_the_format(expected):
    names = _png_names()
    assert len(names) >= 100
    for ctype, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
        for d in depths:
            for i in (0, 1):
                assert any(n.startswith(f"c{ctype}_d{d}_") and n.endswith(f"_i{i}") for n in names)
    assert sum(int(expected[f"png/{n}/err"]) != 0 for n in names) >= 6


@pytest.mark.parametrize("name", _png_names())
def test_png_decoder_equals_reference_lodepng(probe, expected, tmp_path, name):
    err, w, h, rgba = _decode(probe, os.path.join(GOLD, name + ".png"), str(tmp_path / "o.bin"))
    ref_err = int(expected[f"png/{name}/err"])
    assert (err != 0) == (ref_err != 0), f"reference error code {ref_err}, this decoder {'rejects' if err else 'accepts'}"
    if ref_err == 0:
        assert [w, h] == list(expected[f"png/{name}/size"])
        assert np.array_equal(rgba, expected[f"png/{name}/rgba"])


def test_json_numbers_equal_reference_library(probe, expected, tmp_path):
    out = str(tmp_path / "n.bin")
    subprocess.run([probe, "json", os.path.join(GOLD, "numbers.json"), out], check=True, capture_output=True)
    raw = open(out, "rb").read()
    n = struct.unpack("<I", raw[:4])[0]
    fl = np.frombuffer(raw, np.float32, n, 4)
    m = struct.unpack("<I", raw[4 + 4 * n:8 + 4 * n])[0]
    iv = np.frombuffer(raw, np.int32, m, 8 + 4 * n)
    ifl = np.frombuffer(raw, np.float32, m, 8 + 4 * n + 4 * m)
    assert np.array_equal(fl.view(np.uint32), expected["json/floats"].view(np.uint32)), \
        np.nonzero(fl.view(np.uint32) != expected["json/floats"].view(np.uint32))
    assert np.array_equal(iv, expected["json/ints"])
    assert np.array_equal(ifl.view(np.uint32), expected["json/ints_as_float"].view(np.uint32))
    # the quirk the loader depends on: sphere centres and lamp positions truncate toward zero (scene.cpp:97-99)
    doc = open(os.path.join(GOLD, "numbers.json")).read()
    toks = doc.split('"ints": [')[1].split("]")[0].split(", ")
    got = dict(zip(toks, iv))
    assert got["0.7"] == 0 and got["-3.9"] == -3 and got["-0.9"] == 0 and got["2.5"] == 2 and got["-2.5"] == -2


def test_json_reader_rejects_what_the_grammar_rejects(probe, tmp_path):
    for bad in ('{"floats": [01], "ints": []}', '{"floats": [.5], "ints": []}', '{"floats": [1.], "ints": []}',
                '{"floats": [+1], "ints": []}', '{"floats": [1e], "ints": []}', '{"floats": [0x10], "ints": []}',
                '{"floats": [nan], "ints": []}', "[" * 100000):
        p = tmp_path / "bad.json"
        p.write_text(bad)
        r = subprocess.run([probe, "json", str(p), str(tmp_path / "o.bin")], capture_output=True, text=True)
        assert r.returncode == 1 and "json" in r.stderr, bad[:40]


@pytest.mark.skipif(not os.path.exists(REF_CODEC), reason="reference build (oracle/_ref) only exists in the build container")
def test_committed_fixtures_are_what_the_reference_build_says(expected, tmp_path):
    tmp = str(tmp_path / "o.bin")
    for name in _png_names():
        err, w, h, rgba = _decode(REF_CODEC, os.path.join(GOLD, name + ".png"), tmp)
        assert err == int(expected[f"png/{name}/err"]), name
        if err == 0:
            assert [w, h] == list(expected[f"png/{name}/size"]) and np.array_equal(rgba, expected[f"png/{name}/rgba"]), name
    subprocess.run([REF_CODEC, "json", os.path.join(GOLD, "numbers.json"), tmp], check=True)
    raw = open(tmp, "rb").read()
    n = struct.unpack("<I", raw[:4])[0]
    assert np.array_equal(np.frombuffer(raw, np.uint32, n, 4), expected["json/floats"].view(np.uint32))


# ---------------------------------------------------------------------------------------------------------
# framebuffer class: image_bitmap.h compiles on its own, so the reference's class itself wrote tests/golden/ref_bitmap.bin
# ---------------------------------------------------------------------------------------------------------
REF_BITMAP = os.path.join(ROOT, "oracle", "_ref", "ref_bitmap")
BITMAP_GOLD = os.path.join(ROOT, "tests", "golden", "ref_bitmap.bin")


def test_stand_in_framebuffer_equals_the_references_class(tmp_path):
    """The class the drop-in's unit test is compiled against (tests/cpp/mock_reference.h) leaves the same bytes as the
    reference's image_bitmap (image_bitmap.h:11-75) after the same calls: layout, dropped out-of-range pixels, clipped
    copies, clear."""
    exe = str(tmp_path / "stand_in_bitmap")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", '-DBITMAP_STAND_IN="mock_reference.h"', "-I", os.path.join(ROOT, "tests", "cpp"),
                    "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "oracle", "ref_bitmap_harness.cpp")], check=True)
    out = str(tmp_path / "o.bin")
    subprocess.run([exe, out], check=True)
    got, want = open(out, "rb").read(), open(BITMAP_GOLD, "rb").read()
    assert len(want) > 2000 and got == want
    # the layout the device code and hipexec_render_rect assume: float RGB, row-major, data[(y*W + x)*3 + c]
    w, h = struct.unpack("<ii", want[:8])
    first = np.frombuffer(want, np.float32, w * h * 3, 8)
    second = np.frombuffer(want, np.float32, w * h * 3, 8 + 4 * w * h * 3 + 8).reshape(h, w, 3)
    assert (w, h) == (7, 5) and not first.any()
    assert second[3, 2, 0] == 100.0 * 3 + 2 and second[3, 2, 1] == 1.0 and second[3, 2, 2] == -3.0


@pytest.mark.skipif(not os.path.exists(REF_BITMAP), reason="reference build (oracle/_ref) only exists in the build container")
def test_committed_framebuffer_fixture_is_what_the_reference_class_does(tmp_path):
    out, cfg = str(tmp_path / "o.bin"), str(tmp_path / "c.json")
    subprocess.run([REF_BITMAP, out, cfg], check=True)
    assert open(out, "rb").read() == open(BITMAP_GOLD, "rb").read()
    assert open(cfg).read() == open(CONFIG_GOLD).read()


CONFIG_GOLD = os.path.join(ROOT, "tests", "golden", "ref_config.json")


def test_constants_and_eps_comparisons_are_the_references():
    """config.h:8-20 as the reference's own preprocessor and compiler see it (written by oracle/_ref/ref_bitmap): the
    constants this build takes over, and the float-equivalent forms of the comparisons against the double EPS."""
    import json
    import re
    c = json.load(open(CONFIG_GOLD))
    assert c["EPS"] == 0.0001 and c["sizeof_EPS"] == 8                      # a double literal: floats are promoted
    # x < EPS  <=>  x <= 0.0001f   and   x > EPS  <=>  x > 0.0001f   for the floats around 1e-4 (the forms rt_device.hpp and
    # the oracle use, ray_tracer.cpp:191,305,311)
    assert c["eps_float_bits"] == int(np.float32(0.0001).view(np.uint32))
    for lt, gt, le_f in c["around_eps"]:
        assert lt == le_f and gt == 1 - le_f
    hdr = open(os.path.join(ROOT, "include", "hip_executor.h")).read()
    assert int(re.search(r"#define\s+HIPEXEC_AO_RAYS\s+(\d+)", hdr).group(1)) == c["AO_RAYS_COUNT"]
    orc = open(os.path.join(ROOT, "oracle", "rt_oracle.c")).read()
    assert "HIPEXEC_AO_RAYS" in orc                                          # the oracle takes the count from the same header
    # what the reference's default build does, which is what the default flags of this build do
    assert c["GPU_ACCELERATION"] and not c["ENABLE_AO"] and not c["ENABLE_REFLECTION"] and not c["RENDER_PARALLEL"]
    assert (c["WIDTH"], c["HEIGHT"], c["SUB_BLOCK_WIDTH"], c["SUB_BLOCK_HEIGHT"], c["MAX_REFLECTION_DEPTH"]) == (800, 600, 48, 48, 2)
    drv = open(os.path.join(ROOT, "tests", "cpp", "dropin_driver.cpp")).read()
    assert "48" in drv                                                       # the block walk of the drop-in's unit test uses the same block size
