"""The C-ABI shared library loads without a GPU and exports every symbol include/*.h declares;
the ctypes mirrors have the layout the C compiler gives the structs."""
import ctypes as C
import glob
import os
import re
import subprocess

import pytest

from raytracer_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(hipexec_[a-z_0-9]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_abi.LIB_PATH), "build the library first (python __graft_entry__.py)"
    lib = C.CDLL(_abi.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 11
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/ but not exported"
    assert sorted(_abi.PROTOTYPES) == declared, "ctypes prototypes out of sync with the header"
    assert _abi.load_library().hipexec_abi_version() == _abi.ABI_VERSION


def test_struct_layouts_match_the_c_compiler(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hip_executor.h"\nint main(void){'
                   'printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(hipexec_scene_desc), sizeof(hipexec_render_cfg),'
                   'sizeof(hipexec_stats), sizeof(hipexec_hit), offsetof(hipexec_scene_desc, cam_pos),'
                   'offsetof(hipexec_stats, launches), offsetof(hipexec_stats, total_ms));return 0;}')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out == [C.sizeof(_abi.SceneDesc), C.sizeof(_abi.RenderCfg), C.sizeof(_abi.Stats), C.sizeof(_abi.Hit),
                   _abi.SceneDesc.cam_pos.offset, _abi.Stats.launches.offset, _abi.Stats.total_ms.offset]


def test_no_gpu_calls_fail_cleanly_without_a_device():
    """Argument validation happens before any HIP call, so it can be exercised on a CPU-only host."""
    lib = _abi.load_library()
    h = C.c_void_p()
    assert lib.hipexec_create(C.byref(h), None, 0) == -1
    assert b"NULL" in lib.hipexec_last_error()
    assert lib.hipexec_render(None, None, None, None, None) == -1
    assert lib.hipexec_closest_hits(None, 0, None, 0, None) == -1
    lib.hipexec_destroy(None)       # no-op


def test_product_package_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|librt_oracle|rt_oracle\.h|orc_[a-z_]+\(", re.M)
    for path in glob.glob(os.path.join(ROOT, "raytracer_amd", "**", "*.py"), recursive=True):
        assert not pat.search(open(path).read()), f"{path} uses the oracle"
    for path in glob.glob(os.path.join(ROOT, "raytracer_amd", "csrc", "*")):
        assert not pat.search(open(path).read()), f"{path} uses the oracle"


def test_build_flags_keep_the_exact_arithmetic_exact():
    """The exact tests must be compiled without contraction into fused multiply-adds and with correctly rounded division
    and square root (the reference's x86 build rounds after every operation), and without fast-math of any kind."""
    import __graft_entry__ as g
    flags = g.HIPCC_FLAGS
    assert "-ffp-contract=off" in flags
    assert "-fhip-fp32-correctly-rounded-divide-sqrt" in flags
    assert not any(f.startswith("-ffast-math") or f.startswith("-Ofast") or f.startswith("-funsafe-math") or
                   f.startswith("-ffinite-math") or f.startswith("-fno-signed-zeros") for f in flags)
    assert "--offload-arch=gfx950" in flags
