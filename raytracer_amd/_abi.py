"""ctypes mirror of include/hip_executor.h (struct layouts and function prototypes).

The same structures are consumed by the HIP executor (the product) and by the CPU oracle
(test infrastructure), so both see byte-identical scene data.
"""
import ctypes as C
import os

HIPEXEC_MODE_CPU = 0
HIPEXEC_MODE_CL = 1
HIPEXEC_TILE_ROWS = 8
HIPEXEC_FLAG_TIME_KERNELS = 1
HIPEXEC_FLAG_TIME_CLASS_SHIFT = 8
HIPEXEC_K_PRIMARY, HIPEXEC_K_CLOSEST, HIPEXEC_K_ANYHIT, HIPEXEC_K_SHADE = 0, 1, 2, 3
HIPEXEC_FLAG_SCAN = 2
HIPEXEC_FLAG_BVH = 4
HIPEXEC_FLAG_BILINEAR = 8
HIPEXEC_FLAG_REFRACT = 16
HIPEXEC_FLAG_AO = 32
HIPEXEC_ALGO_SCAN, HIPEXEC_ALGO_BVH = 0, 1
K_PRIMARY, K_CLOSEST, K_ANYHIT, K_SHADE, K_COUNT = 0, 1, 2, 3, 4
ABI_VERSION = 7
STATIC_MAX_RAYS = 1 << 18

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_tri", C.c_uint32), ("tri_p", _fp), ("tri_e1", _fp), ("tri_e2", _fp), ("tri_norm", _fp),
        ("tri_uv", _fp), ("tri_mat", _ip),
        ("n_sph", C.c_uint32), ("sph_center", _fp), ("sph_radius", _fp), ("sph_mat", _ip),
        ("n_mat", C.c_uint32), ("mat_color", _fp), ("mat_kd", _fp), ("mat_ks", _fp), ("mat_hard", _fp),
        ("mat_kr", _fp), ("mat_tex", _ip),
        ("n_tex", C.c_uint32), ("tex_rgba", C.POINTER(C.c_uint8)), ("tex_offset", C.POINTER(C.c_uint64)),
        ("tex_w", C.POINTER(C.c_uint32)), ("tex_h", C.POINTER(C.c_uint32)),
        ("n_lamp", C.c_uint32), ("lamp_pos", _fp), ("lamp_intensity", _fp), ("lamp_distance", _fp),
        ("cam_pos", C.c_float * 3), ("cam_mat", C.c_float * 9),
        ("horizon", C.c_float * 3), ("ambient", C.c_float * 3),
        ("mat_kt", _fp), ("mat_ior", _fp), ("ambient_factor", C.c_float),
    ]


class RenderCfg(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("mode", C.c_int32), ("max_depth", C.c_int32),
        ("tone", C.c_int32), ("tile_first", C.c_int32), ("tile_step", C.c_int32), ("flags", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64), ("rays_bounce", C.c_uint64), ("rays_shadow", C.c_uint64),
        ("tests_tri_closest", C.c_uint64), ("tests_sph_closest", C.c_uint64),
        ("tests_tri_any", C.c_uint64), ("tests_sph_any", C.c_uint64),
        ("tests_tri_primary", C.c_uint64), ("exact_groups", C.c_uint64 * 3), ("scanned_chunks", C.c_uint64 * 3),
        ("bvh_nodes", C.c_uint64 * 3), ("bvh_tests", C.c_uint64 * 3), ("bvh_stack_overflows", C.c_uint64), ("algorithm", C.c_uint32), ("lanes", C.c_uint32),
        ("launches", C.c_uint32 * K_COUNT), ("kernel_ms", C.c_float * K_COUNT), ("total_ms", C.c_float), ("retraced", C.c_uint32),
    ]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k in (
            "rays_primary", "rays_bounce", "rays_shadow", "tests_tri_closest", "tests_sph_closest",
            "tests_tri_any", "tests_sph_any", "tests_tri_primary")}
        d["exact_groups"] = [int(x) for x in self.exact_groups]
        d["scanned_chunks"] = [int(x) for x in self.scanned_chunks]
        d["bvh_nodes"] = [int(x) for x in self.bvh_nodes]
        d["bvh_tests"] = [int(x) for x in self.bvh_tests]
        d["bvh_stack_overflows"] = int(self.bvh_stack_overflows)
        d["launches"] = list(self.launches)
        d["kernel_ms"] = [float(x) for x in self.kernel_ms]
        d["total_ms"] = float(self.total_ms)
        d["retraced"] = int(self.retraced)
        return d

    @property
    def rays(self):
        return int(self.rays_primary + self.rays_bounce + self.rays_shadow)


class Hit(C.Structure):
    _fields_ = [
        ("id", C.c_int32), ("mat", C.c_int32), ("t", C.c_float), ("u", C.c_float), ("v", C.c_float),
        ("point", C.c_float * 3), ("norm", C.c_float * 3), ("color", C.c_float * 3),
    ]


# numpy structured dtype with the same layout as Hit
import numpy as _np  # noqa: E402

HIT_DTYPE = _np.dtype([
    ("id", "<i4"), ("mat", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4"),
    ("point", "<f4", 3), ("norm", "<f4", 3), ("color", "<f4", 3)])
assert HIT_DTYPE.itemsize == C.sizeof(Hit)

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(REPO_ROOT, "raytracer_amd", "lib", "libhip_executor.so")

# name -> (restype, argtypes); every symbol include/hip_executor.h declares
ROWS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32)

PROTOTYPES = {
    "hipexec_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(SceneDesc), C.c_int]),
    "hipexec_destroy": (None, [C.c_void_p]),
    "hipexec_clone": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    "hipexec_set_camera": (C.c_int, [C.c_void_p, _fp, _fp]),
    "hipexec_optimize": (C.c_int, [C.c_void_p, _fp]),
    "hipexec_trim": (C.c_int, [C.c_int]),
    "hipexec_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hipexec_set_lanes": (C.c_int, [C.c_void_p, C.c_int]),
    "hipexec_set_algorithm": (C.c_int, [C.c_void_p, C.c_int]),
    "hipexec_render": (C.c_int, [C.c_void_p, C.POINTER(RenderCfg), C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "hipexec_render_progressive": (C.c_int, [C.c_void_p, C.POINTER(RenderCfg), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(Stats)]),
    "hipexec_render_rect": (C.c_int, [C.c_void_p, C.POINTER(RenderCfg), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                            C.POINTER(Stats)]),
    "hipexec_render_device": (C.c_int, [C.c_void_p, C.POINTER(RenderCfg), C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "hipexec_sync": (C.c_int, [C.c_void_p]),
    "hipexec_closest_hits": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]),
    "hipexec_any_hits": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]),
    "hipexec_any_hits_static": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]),
    "hipexec_trace_rays": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Stats)]),
    "hipexec_last_error": (C.c_char_p, []),
    "hipexec_abi_version": (C.c_int, []),
}

_lib = None


def load_library(path=None):
    """Load the HIP executor.  There is no fallback: a missing library is an error."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"HIP executor library not found at {p}; build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)")
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.hipexec_abi_version() != ABI_VERSION:
        raise RuntimeError("libhip_executor.so ABI version mismatch; rebuild")
    if path is None:
        _lib = lib
    return lib
